"""Idealised bound: strided first round + phase-2 claims by predicted cost (max of the running cost estimates of the nearest first-round chunks in the same image column)."""
import sys, time
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests'); sys.path.insert(0,'/root/repo/tests/dev')
import numpy as np
import online_order_sim as S
from oracle import pyoracle as po
import raycore_jl_amd as rc
from helpers import build_oracle
W, LANES, POOL, REFILL = S.W, S.LANES, S.POOL, S.REFILL

def simulate2(cost, row_chunks, stride, scale, dt=0.1, tau=(0.18,0.08), oracle_pred=False, cmax=None):
    n=len(cost); nb=(n+POOL-1)//POOL
    rows=nb//row_chunks
    first_rows=np.arange(0,rows,stride)
    first=(first_rows[:,None]*row_chunks+np.arange(row_chunks)[None,:]).reshape(-1)
    is_first=np.zeros(nb,bool); is_first[first]=True
    est=np.zeros(nb,np.int32)           # running cost estimate of a chunk (longest finished lifetime / oldest in-flight age seen at a refill)
    claimed=np.zeros(nb,bool)
    # nearest first-round rows above / below for each chunk
    r_of=np.arange(nb)//row_chunks; c_of=np.arange(nb)%row_chunks
    up=(r_of//stride)*stride; dn=np.minimum(up+stride,(rows-1)//stride*stride)
    nb_up=up*row_chunks+c_of; nb_dn=dn*row_chunks+c_of
    rem=np.zeros((W,LANES),np.int32); life=np.zeros((W,LANES),np.int32); lane_ray=np.full((W,LANES),-1,np.int64)
    pool_next=np.zeros(W,np.int64); pool_end=np.zeros(W,np.int64); exhausted=np.zeros(W,bool); alive=np.ones(W,bool); prog=np.zeros(W)
    simd=(np.arange(W)//12%256)*4+(np.arange(W)%12)%4
    nf=[0]; n_left=[nb]
    def next_range(w):
        if nf[0]<len(first):
            c=first[nf[0]]; nf[0]+=1
        else:
            if n_left[0]==0: return None
            if oracle_pred: pred=cmax.astype(np.int64).copy()
            else: pred=np.maximum(est[nb_up],est[nb_dn]).astype(np.int64)
            pred[claimed]=-1
            c=int(np.argmax(pred))
            if claimed[c]: return None
        claimed[c]=True; n_left[0]-=1
        a=c*POOL; return a,min(a+POOL,n)
    def refill(w):
        free=np.nonzero(rem[w]==0)[0]
        done=free[lane_ray[w,free]>=0]
        if len(done):
            ch=lane_ray[w,done]//POOL
            np.maximum.at(est,ch,life[w,done])
        liv=np.nonzero(rem[w]>0)[0]
        if len(liv):
            np.maximum.at(est,lane_ray[w,liv]//POOL,life[w,liv])
        lane_ray[w,free]=-1
        i=0
        while i<len(free):
            if pool_next[w]==pool_end[w]:
                if exhausted[w]: break
                r=next_range(w)
                if r is None: exhausted[w]=True; break
                pool_next[w],pool_end[w]=r
            k=int(min(len(free)-i,pool_end[w]-pool_next[w]))
            ids=np.arange(pool_next[w],pool_next[w]+k)
            rem[w,free[i:i+k]]=cost[ids]; life[w,free[i:i+k]]=0; lane_ray[w,free[i:i+k]]=ids
            pool_next[w]+=k; i+=k
    for w in range(W): refill(w)
    t=0.0; a_,b_=tau[0]*scale,tau[1]*scale
    while alive.any():
        k=np.bincount(simd[alive],minlength=1024)
        prog[alive]+=dt/(a_+b_*k[simd[alive]])
        step=np.nonzero(alive&(prog>=1.0))[0]; t+=dt
        if len(step)==0: continue
        prog[step]-=1.0
        n_live=(rem[step]>0).sum(axis=1)
        can=~(exhausted[step]&(pool_next[step]==pool_end[step]))
        do=(can&(LANES-n_live>=REFILL))|(n_live==0)
        adv=step[~do]; lm=rem[adv]>0; rem[adv]-=lm; life[adv]+=lm
        for w in step[do]:
            if not can[np.searchsorted(step,w)]: alive[w]=False; continue
            refill(w)
            if (rem[w]>0).sum()==0 and exhausted[w] and pool_next[w]==pool_end[w]: alive[w]=False
    return t

sc=rc.scenes; cfg=sc.config_c3(); o=build_oracle(po,cfg)
for res,meas in ((2048,505.0),(1024,226.0)):
    rays=sc.c3_primary_rays(cfg,res,res)
    _,cnt=o.trace(rays,nthreads=8,counters=True)
    cost=np.maximum((cnt[:,0].astype(np.int64)+2*cnt[:,1]).astype(np.int32),1)
    nb=len(cost)//128; cmax=cost.reshape(nb,128).max(axis=1); rc_=res//128
    S.TAPER=0   # (the coarse scheme deals whole chunks: compare against natural order without the taper as well as with it)
    t0=S.simulate(cost,dt=0.1); 
    S.TAPER=12
    t12=S.simulate(cost,dt=0.1); scale=meas/t12
    S.TAPER=0; t_nat0=S.simulate(cost,scale=scale,dt=0.1); S.TAPER=12; t_nat=S.simulate(cost,scale=scale,dt=0.1)
    lpt=np.argsort(-cmax,kind='stable'); t_lpt=S.simulate(cost,order=lpt,scale=scale,dt=0.1)
    print(f"C3 {res}x{res}: natural taper12 {t_nat:.1f} us, natural taper0 {t_nat0:.1f}, LPT (taper12) {t_lpt:.1f} ({100*(t_nat/t_lpt-1):+.1f} %)",flush=True)
    rows=nb//rc_
    for stride in (max(2,int(np.ceil(rows/(6144/rc_)))), 8, 16):
        tt=simulate2(cost,rc_,stride,scale)
        tp=simulate2(cost,rc_,stride,scale,oracle_pred=True,cmax=cmax)
        print(f"   strided first round (every {stride}th row) + neighbour-predicted LPT: {tt:.1f} us ({100*(t_nat/tt-1):+.1f} % vs natural taper12, before the ~1-2 % a strided order costs); with a perfect predictor {tp:.1f} ({100*(t_nat/tp-1):+.1f} %)",flush=True)
