#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r04camp; mkdir -p $O
timeout 3000 python3 tools/full_parity_campaign.py > $O/full_parity.log 2>&1; tail -8 $O/full_parity.log
RC_BENCH_FORCE_DIST=1 timeout 900 python3 bench.py --steps 5 --warmup 2 > $O/bench_force_dist.json 2> $O/bench_force_dist.err; tail -c 300 $O/bench_force_dist.err; python3 -c "
import json; d=json.loads(open('$O/bench_force_dist.json').read().strip().splitlines()[-1]); v=d['extras']['view_factors_c5']; print('force_dist', d['value'], {k:(v[k] if not isinstance(v[k],dict) else {x:v[k].get(x) for x in ('seconds','counted','equals_one_gpu')}) for k in ('rows_sharded','rows','rays','totals_rays_sharded','rccl_ranks','one_gpu_count')}, [k for k in d['extras'] if k.endswith('_error')])"
RC_BENCH_FORCE_MULTI=2 timeout 900 python3 bench.py --steps 5 --warmup 2 > $O/bench_force_multi.json 2> $O/bench_force_multi.err; tail -c 300 $O/bench_force_multi.err; python3 -c "
import json; d=json.loads(open('$O/bench_force_multi.json').read().strip().splitlines()[-1]); print('force_multi', json.dumps(d['extras'].get('one_process_multi_device'))[:1500], [k for k in d['extras'] if k.endswith('_error')])"
