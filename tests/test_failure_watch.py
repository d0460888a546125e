"""FailureWatch (gpu-spmv_amd/pagerank_dist.py) against an in-memory store: what ADVICE r03 found — a rank must not be
taken down by its OWN report, a report against one loop must not fail the loops created after it, and several live
loops share one polling thread.  (The cross-process behaviour is in tests/test_distributed_gloo.py.)"""
import importlib
import threading
import time

import pytest

prd = importlib.import_module("gpu-spmv_amd.pagerank_dist")


class MemoryStore:
    def __init__(self):
        self.data, self.lock = {}, threading.Lock()

    def set(self, key, value):
        with self.lock:
            self.data[key] = value.encode() if isinstance(value, str) else value

    def get(self, key):
        with self.lock:
            return self.data[key]

    def check(self, keys):
        with self.lock:
            return all(k in self.data for k in keys)

    def delete_key(self, key):
        with self.lock:
            return self.data.pop(key, None) is not None


@pytest.fixture
def exits(monkeypatch):
    """os._exit inside the watch is recorded, not taken."""
    seen = []
    import os
    monkeypatch.setattr(os, "_exit", lambda code: seen.append(code))
    return seen


def pair(store, **kw):
    """the two ranks' watches of ONE loop (same generation on both sides, as when ranks create their loops in lockstep)"""
    a = prd.FailureWatch(0, 2, store=store, **kw)
    prd.FailureWatch._generation -= 1
    b = prd.FailureWatch(1, 2, store=store, **kw)
    assert a.generation == b.generation
    return a, b


def test_a_rank_is_not_taken_down_by_its_own_report(exits):
    store = MemoryStore()
    w0, w1 = pair(store, poll=0.02, grace=0.1)
    w0.report(RuntimeError("step 3 failed"))          # rank 0 reports, handles the exception, carries on
    time.sleep(0.5)
    w0.check()                                        # nothing to raise: its own story is not a peer's
    assert w0.peer_failed is None
    # ... while rank 1 does hear about it, and — blocked (never calling check) — is taken down after the grace period
    assert w1.peer_failed is not None and "rank 0" in w1.peer_failed
    assert exits == [prd.EXIT_PEER_FAILED] * len(exits) and len(exits) >= 1
    w0.stop(); w1.stop()


def test_a_main_thread_that_notices_in_time_leaves_by_itself(exits):
    store = MemoryStore()
    w0, w1 = pair(store, poll=0.02, grace=0.5)
    w1.report(ValueError("bad shard"))
    deadline = time.time() + 2
    while w0.peer_failed is None and time.time() < deadline:
        time.sleep(0.01)
    with pytest.raises(prd.PeerFailure):
        w0.check()
    time.sleep(0.8)
    assert exits == []                                # acknowledged: no os._exit behind it
    w0.stop(); w1.stop()


def test_a_report_against_one_loop_does_not_fail_the_next_one(exits):
    store = MemoryStore()
    w0, w1 = pair(store, poll=0.02, grace=0.2)
    w1.report(RuntimeError("trial aborted"))
    time.sleep(0.1)
    with pytest.raises(prd.PeerFailure):
        w0.check()
    w0.stop(); w1.stop()                              # the trial's loops are closed
    n0, n1 = pair(store, poll=0.02, grace=0.2)        # the next loop of the same job
    assert n0.generation == w0.generation + 1
    time.sleep(0.6)
    n0.check(); n1.check()
    assert n0.peer_failed is None and n1.peer_failed is None and exits == []
    n0.stop(); n1.stop()


def test_live_watches_share_one_polling_thread_and_it_ends_with_the_last(exits):
    store = MemoryStore()
    watches = [prd.FailureWatch(0, 2, store=store, poll=0.02) for _ in range(4)]
    time.sleep(0.1)
    names = [t.name for t in threading.enumerate() if t.name == "spmv-failure-watch"]
    assert len(names) == 1
    for w in watches:
        w.stop()
    time.sleep(0.3)
    assert not [t for t in threading.enumerate() if t.name == "spmv-failure-watch"]
