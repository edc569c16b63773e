"""CPU: the promises of the batch-slot state machine, checked on its Python restatement (tests/order_model.py); the device is held against the
same restatement word for word in tests/test_gpu_order_model.py."""
import copy

import order_model as om


def run(script, mut=None):
    h = om.History(mut)
    return [h.launch(b, near) for b, near in script], h


def test_a_repeating_batch_has_an_order_by_its_third_launch_and_records_2_to_4_then_about_one_in_eight():
    tr, h = run([(0, ())] * 40)
    assert all(t["skip_left"] == 0 for t in tr)
    assert [t["order_valid"] for t in tr[:4]] == [0, 0, 1, 1] and all(t["order_valid"] for t in tr[2:])
    assert [i + 1 for i, t in enumerate(tr) if t["records"]] == [2, 3, 4, 8, 15, 23, 32, 39]   # the host's cadence: 7, 8, 9, 7, ... launches apart
    assert all(t["streak"] == 0 for t in tr[1:]) and tr[0]["streak"] == 1 and tr[0]["fresh"] == 1
    # a recording waits exactly one launch for its rebuild ...
    assert [i + 1 for i, t in enumerate(tr) if t["pending"][t["sel"]]] == [2, 3, 4, 8, 15, 23, 32, 39]
    # ... and the rebuild pair runs in front of the shape's first launches (2-7: the start-up, and the credit the first, unmatched launch leaves)
    # and after recordings only (9, 16, 24, 33, 40): 11 of 40 launches, 5 of the 33 after the start-up
    assert h.rebuilds == 11


def test_two_alternating_batches_each_learn_from_their_own_launches():
    tr, h = run([(k % 2, ()) for k in range(64)])
    a, b = tr[0::2], tr[1::2]
    assert len({t["sel"] for t in a}) == 1 and len({t["sel"] for t in b}) == 1 and a[0]["sel"] != b[0]["sel"]
    for side in (a, b):
        assert [t["gen"][t["sel"]] for t in side] == list(range(1, 33)) and [t["order_valid"] for t in side][:4] == [0, 0, 1, 1]
        assert [t["records"] for t in side][1:4] == [1, 1, 1]
        assert any(t["records"] and t["gen"][t["sel"]] >= 5 for t in side)   # the host's cadence (7, 8, 9 launches apart: no fixed parity) reaches both batches


def test_five_batches_on_four_slots_never_match_and_the_host_pauses_after_eight():
    tr, _ = run(om.scripts()["A B C D E rotation (five batches, four slots)"])
    assert all(t["fresh"] == 1 and t["order_valid"] == 0 and t["records"] == 0 for t in tr[:8])   # a batch seen for the first time never records
    assert [t["streak"] for t in tr[:8]] == list(range(1, 8)) + [0] and tr[7]["skip_left"] == 64   # the eighth starts the pause (and the count over)
    assert [t["skip_left"] for t in tr[8:]] == list(range(63, 56, -1)) and all(t["clock"] == 8 for t in tr[8:])   # inside the pause nothing else moves


def test_a_moving_camera_starts_over_in_its_slot_every_frame_and_the_pause_ends():
    tr, _ = run(om.scripts()["still, then a moving camera, then still again"])
    moving = tr[4:16]
    # matched (the still batch's slot follows the camera: nobody else is evicted) but never a repeat: natural order, nothing recorded, from the first frame on
    assert all(t["fresh"] == 1 and t["order_valid"] == 0 and t["records"] == 0 and t["sel"] == tr[3]["sel"] for t in moving[:8])
    assert [t["streak"] for t in moving[:8]] == list(range(1, 8)) + [0] and all(t["clock"] == 12 and t["skip_left"] > 0 for t in moving[8:])
    assert all(t["clock"] == 12 and t["skip_left"] > 0 for t in tr[16:])                  # still inside the pause
    long, _ = run(om.scripts()["never repeating, the pause, then a repeating batch"])
    assert [t["clock"] for t in long[:8]] == list(range(1, 9)) and [t["skip_left"] for t in long[8:72]] == list(range(63, -1, -1)) and all(t["clock"] == 8 for t in long[8:72])
    back = long[72:]
    assert back[0]["fresh"] == 1 and back[0]["streak"] == 1 and [t["streak"] for t in back[1:]] == [0] * 5 and [t["order_valid"] for t in back] == [0, 0, 1, 1, 1, 1]


def test_mutants_of_the_transitions_change_the_trace():
    """what the device comparison relies on: each of these one-token changes of the state machine shows up in the modelled words of at least
    one script (so the same change made in order_select / order_commit / rc_cost_order_setup fails tests/test_gpu_order_model.py)"""
    ref = {name: run(sc)[0] for name, sc in om.scripts().items()}
    for field, value in (("record_first", 1), ("record_last", 3), ("first_cadence_record", 9), ("repeats_only", False), ("only_repeats_continue", False), ("rebuild_on_pending_word", False), ("give_up_after", 9)):
        m = copy.copy(om.Mutations())
        setattr(m, field, value)
        assert any(run(sc, m)[0] != ref[name] for name, sc in om.scripts().items()), field
