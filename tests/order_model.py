"""Python restatement of the batch-slot state machine of cost-ordered claiming (test infrastructure).

The product side is split between the host (rc_cost_order_setup, raycore.jl_amd/csrc/rc_traverse.hip: when a batch is asked to record, when
the rebuild kernels run) and the device (order_select / order_commit, rc_traverse_core.h: which of the
history's four batch slots a launch belongs to, what the launch does with it, and the pause of a shape whose batches do not repeat --
evaluated INSIDE the launch since round 5; k_order_scatter:
a recording becomes an order).  A wrong transition cannot change a hit -- the claim order is result-neutral -- but it can silently turn the
feature into overhead, so the transitions are restated here, driven with scripted launch sequences, and compared word for word with the
header words the device leaves behind (tests/test_gpu_order_model.py); tests/test_order_model.py checks the restatement's own promises on
the CPU.

A launch is described by what order_select can see of it: `batch` (any hashable: launches with equal ids trace identical rays) and `near`
(a set of batch ids whose sample rays are within the matching threshold of this launch's without being identical: the frame before of a
moving camera).  Data-dependent words (the reporting threshold and the scale of the cost classes) are outside the model: it says WHETHER a
launch records, not with which threshold.  The model assumes the caller waits for every launch (the pinned words the host reads are then
current); a caller that enqueues far ahead sees them late, which delays a rebuild and changes no result -- except for the pause count, which
used to keep such a caller "paused" on the host for as long as it did not wait: since round 6 the word carries the number of the launch that
wrote it and the host takes one launch off for every launch enqueued since (tests/test_gpu_order_model.py::
test_a_caller_that_enqueues_far_ahead_still_gets_its_order).
"""

K_SLOTS = 4                    # kHistSlots
GIVE_UP_AFTER, GIVE_UP_FOR = 8, 64   # kGiveUpAfter / kGiveUpFor (rc_traverse_core.h): counted and decided on the device
EARLY_LAUNCHES, CREDIT_AFTER_NON_REPEAT = 6, 6

# header words of the history (rc_traverse_core.h kHist*)
SEL, ORDER_VALID, LIFE_THR, CLOCK, FRESH, RECORDED = 0, 1, 2, 3, 4, 5
STAMP, GEN, PENDING, STREAK, SKIP_LEFT, HAS_ORDER = 8, 12, 32, 36, 38, 40


class Mutations:
    """knobs for the mutants the comparison has to catch"""
    record_first, record_last = 2, 4   # a slot records its launches 2-4 ...
    first_cadence_record = 8           # ... and then when the host's cadence asks (launches 8, 15, 23, 32, 39, ... of the shape)
    repeats_only = True                # the streak counts launches that are not exact repeats (False: only launches that matched no slot at all)
    only_repeats_continue = True       # a launch that matches a slot without repeating its batch starts over in it (False: it continues the slot's history, the rule before)
    rebuild_on_pending_word = True     # the host runs the rebuild kernels when the device reported a waiting recording
    give_up_after = GIVE_UP_AFTER      # consecutive non-repeats that start a pause


class History:
    """One launch shape's history: device header + the host's logic.  launch() returns the header words the device holds after the launch (a
    dict of the modelled words); a launch inside a pause changes none of them but `skip_left`."""

    def __init__(self, mut=None):
        self.m = mut or Mutations()
        self.clock = 0
        self.stamp = [0] * K_SLOTS
        self.gen = [0] * K_SLOTS
        self.pending = [0] * K_SLOTS
        self.has_order = [0] * K_SLOTS
        self.batch = [None] * K_SLOTS      # whose sample rays the slot holds
        self.streak = 0                    # kHistFreshStreak
        self.skip_left = 0                 # kHistSkipLeft: launches of a pause still to go
        self.host_streak = 0               # pinned word 0 as the host last saw it
        self.host_pending = 0              # pinned word 1
        self.host_skip = 0                 # pinned word 2
        self.host_gen = 0                  # launches of the shape
        self.credit = 0
        self.next_record = self.m.first_cadence_record
        self.records_asked = 0
        self.rebuilds = 0                  # how often the rebuild pair was enqueued (not a device word)
        self.last = {"sel": 0, "order_valid": 0, "records": 0, "clock": 0, "fresh": 0, "stamp": [0] * K_SLOTS, "gen": [0] * K_SLOTS,
                     "pending": [0] * K_SLOTS, "has_order": [0] * K_SLOTS, "streak": 0, "skip_left": 0}

    def launch(self, batch, near=()):
        # ---- host: rc_cost_order_setup (when the rebuild pair runs, when a batch is asked to record) ----
        self.host_gen += 1
        paused = self.host_skip > 1        # this launch and the next are inside the pause: nothing records, nothing to rebuild
        if self.host_streak > 0:
            self.credit = CREDIT_AFTER_NON_REPEAT
        if paused:
            self.credit = 0
        if self.host_gen >= 2 and not paused and (self.host_gen <= EARLY_LAUNCHES or self.credit > 0 or (self.m.rebuild_on_pending_word and self.host_pending)):
            self.rebuilds += 1             # k_order_count + k_order_scatter: every waiting recording becomes an order
            for k in range(K_SLOTS):
                if self.pending[k]:
                    self.pending[k], self.has_order[k] = 0, 1
            self.host_pending = 0
        if self.credit > 0:
            self.credit -= 1
        want_record = 0
        if self.host_gen >= self.next_record and not paused:
            want_record = 1
            self.next_record = self.host_gen + 7 + self.records_asked % 3
            self.records_asked += 1
            self.credit = max(self.credit, 1)
        # ---- device: a launch inside a pause does not look at its rays ----
        if self.skip_left > 0:
            self.skip_left -= 1
            self.host_skip = self.skip_left
            self.last = dict(self.last, skip_left=self.skip_left, pending=list(self.pending), has_order=list(self.has_order))
            return self.last
        # ---- device: order_select, evaluated by the launch itself ----
        best, exact = None, False
        for k in range(K_SLOTS):           # the closest slot below the threshold; an identical batch has distance 0 and wins
            if self.stamp[k] == 0:
                continue
            if self.batch[k] == batch:
                best, exact = k, True
                break
            if best is None and self.batch[k] in near:
                best = k
        # only a REPEAT continues a slot's history; a launch that merely resembles a remembered batch starts over in that batch's slot
        fresh = (not exact) if self.m.only_repeats_continue else best is None
        sel = best if best is not None else min(range(K_SLOTS), key=lambda k: (self.stamp[k], k))   # the matched slot, else an empty one, else the least recently used
        gen = 1 if fresh else self.gen[sel] + 1
        record = (self.m.record_first <= gen <= self.m.record_last) or (want_record and gen >= 5)
        valid = (not fresh) and self.has_order[sel] != 0
        # ---- device: order_commit, by one wave of the launch (into the header's other copy) ----
        self.batch[sel] = batch            # the slot's samples follow the batch
        self.clock += 1
        self.stamp[sel] = self.clock
        self.gen[sel] = gen
        if fresh:
            self.pending[sel] = 0
            self.has_order[sel] = 0
        if record:
            self.pending[sel] = 1
        not_counted = exact if self.m.repeats_only else best is not None
        self.streak = 0 if not_counted else self.streak + 1
        if self.streak >= self.m.give_up_after:   # the shape's batches do not repeat: its next launches go out outside the mechanism
            self.streak, self.skip_left = 0, GIVE_UP_FOR
        self.host_streak = self.streak     # (the test waits for every launch: the pinned words are current at the next one)
        self.host_pending = int(any(self.pending))
        self.host_skip = self.skip_left
        self.last = {"sel": sel, "order_valid": int(valid), "records": int(record), "clock": self.clock, "fresh": int(fresh),
                     "stamp": list(self.stamp), "gen": list(self.gen), "pending": list(self.pending), "has_order": list(self.has_order), "streak": self.streak,
                     "skip_left": self.skip_left}
        return self.last


def device_words(w):
    """the modelled words out of a dump of the first 48 header words (numpy uint32)"""
    return {"sel": int(w[SEL]), "order_valid": int(w[ORDER_VALID]), "records": int(w[RECORDED]), "clock": int(w[CLOCK]), "fresh": int(w[FRESH]),
            "stamp": [int(x) for x in w[STAMP:STAMP + K_SLOTS]], "gen": [int(x) for x in w[GEN:GEN + K_SLOTS]],
            "pending": [int(x) for x in w[PENDING:PENDING + K_SLOTS]], "has_order": [int(x) for x in w[HAS_ORDER:HAS_ORDER + K_SLOTS]], "streak": int(w[STREAK]), "skip_left": int(w[SKIP_LEFT])}


def scripts():
    """name -> list of (batch, near) launches.  Batches are identified by the index of a camera position; ("m", f) = frame f of a camera that
    moves a little per frame from position 0 (frame f is near frame f - 1, and frame 1 is near position 0)."""
    def still(b, n):
        return [(b, ())] * n

    def moving(first, n):
        return [(("m", f), ((("m", f - 1),) if f > 1 else (0,))) for f in range(first, first + n)]
    return {
        "A x 20": still(0, 20),
        "A B A B": [(k % 2, ()) for k in range(24)],
        "A B C D E rotation (five batches, four slots)": [(k % 5, ()) for k in range(15)],
        "still, then a moving camera, then still again": still(0, 4) + moving(1, 12) + [(0, (("m", 12),))] + still(0, 3),   # (position 0 is still within the threshold of where the camera stopped)
        "never repeating, the pause, then a repeating batch": [(100 + k, ()) for k in range(10)] + [(200 + k, ()) for k in range(62)] + still(3, 6),
        "A A B B B A A A A": still(0, 2) + still(1, 3) + still(0, 4),
        "A B C rotation (three batches, each with a slot)": [(k % 3, ()) for k in range(21)],   # their launches 4 record late: only the device's "a recording waits" word brings the rebuild
    }
