#!/usr/bin/env python3
"""Request coalescing from several threads at once: submitters keep queueing while another thread waits for / downloads the
batch in flight (the coalescer's lock is released for that), a flusher launches batches on a timer; every ticket gets the
bytes its request gets alone (the oracle's per-call result).  Run with ERLAMSA_HIP_LIB=<emu lib>."""
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import pyoracle as po
import util
import erlamsa_amd as ea

MUTS, PATS = "bd,bf,sr,num,ld,lr", "od,nd,bu"
NT, PER = 4, 24
inputs = util.corpus_mixed(NT * PER, 300, seed=21)
seeds = np.array([(1000 + i, 3 * i + 1, 7 * i + 2) for i in range(NT * PER)], dtype=np.int64)
dq, oq = po.pack(inputs)
want, wst, _, _ = po.fuzz_batch(dq, oq, seeds=seeds, mutations=MUTS, patterns=PATS)
eng = ea.Engine(0)
eng.configure(mutations=MUTS, patterns=PATS)
eng.coalesce_limits(16, 1 << 20)
stop = threading.Event()
errors = []


def flusher():
    while not stop.is_set():
        try:
            eng.flush()
        except Exception as ex:                       # noqa: BLE001
            errors.append(("flush", repr(ex)))
        time.sleep(0.002)


def client(t):
    try:
        mine = list(range(t * PER, (t + 1) * PER))
        tickets = {}
        for i in mine:
            tickets[i] = eng.submit(inputs[i], tuple(int(x) for x in seeds[i]))
            if i % 5 == 0:                               # poll some early, while others are still submitting
                j = mine[0] + (i - mine[0]) // 2
                if j in tickets:
                    r = eng.poll(tickets[j])
                    if r is not None:
                        assert r == (int(wst[j]), want[j]), j
                        del tickets[j]
        deadline = time.time() + 120
        while tickets and time.time() < deadline:
            for i in list(tickets):
                r = eng.poll(tickets[i])
                if r is not None:
                    assert r == (int(wst[i]), want[i]), i
                    del tickets[i]
            time.sleep(0.001)
        assert not tickets, "tickets never served: %s" % sorted(tickets)[:5]
    except Exception as ex:                           # noqa: BLE001
        errors.append((t, repr(ex)))


fl = threading.Thread(target=flusher)
fl.start()
cs = [threading.Thread(target=client, args=(t,)) for t in range(NT)]
[c.start() for c in cs]
[c.join() for c in cs]
stop.set()
fl.join()
eng.close()
assert not errors, errors[:3]
print("coalescing from %d threads ok: %d requests" % (NT, NT * PER))
