"""Two engines on one GPU own their state (include/pdftable_hip.h, conventions): interleaving their calls gives each
the results it gives alone -- bit-exact, the kernels are deterministic.  Covers the engine-owned scratch that is NOT in
the per-call arena: the LSTM exchange buffers of the recognizer and the decode state of pt_tsr_forward_decode."""
import numpy as np
import pytest
import torch

from pdf_table_amd import lib as L
from pdf_table_amd.synth_weights import crnn_state_dict, lore_dla34_state_dict
from pdf_table_amd.weights import pack_crnn, pack_lore_dla34

pytestmark = pytest.mark.gpu


def _engine(seed):
    from pdf_table_amd.engine import HipEngine
    e = HipEngine(0)
    e.load_weights(L.PT_MODEL_CRNN, pack_crnn(crnn_state_dict(seed=seed)))
    e.load_weights(L.PT_MODEL_LORE_DLA34, pack_lore_dla34(lore_dla34_state_dict(seed=seed, hm_bias=(-1.2, -0.6))))
    return e


def test_two_engines_interleaved_equal_each_alone():
    g = torch.Generator().manual_seed(5)
    gray = (torch.rand(200, L.PT_REC_H, L.PT_REC_W, generator=g) * 2 - 1).to(torch.bfloat16).cuda()
    x4 = torch.zeros(2, 256, 256, 4)
    x4[..., :3] = torch.randn(2, 256, 256, 3, generator=g) * 0.7
    x4 = x4.to(torch.bfloat16).cuda()

    def run(e):
        ids, mx = e.rec_forward_net(gray)
        c, d, l = e.tsr_forward_decode(x4, wiz_rev=True, vis_thresh=0.2, sync=True)
        e.check()
        return ids.cpu().numpy(), mx.cpu().numpy(), np.asarray(c), d.cpu().numpy(), l.cpu().numpy()

    alone = []
    for seed in (3, 4):
        e = _engine(seed)
        alone.append(run(e))
        e.close()
    assert not np.array_equal(alone[0][0], alone[1][0])          # different weights: different token ids
    a, b = _engine(3), _engine(4)
    try:
        ia, _ = a.rec_forward_net(gray)                          # a and b alternate, results fetched at the end
        ib, _ = b.rec_forward_net(gray)
        ca, da, la = a.tsr_forward_decode(x4, wiz_rev=True, vis_thresh=0.2, sync=False)
        cb, db, lb = b.tsr_forward_decode(x4, wiz_rev=True, vis_thresh=0.2, sync=False)
        ia2, ma2 = a.rec_forward_net(gray)
        torch.cuda.synchronize()
        a.check()
        b.check()
        got_a = run(a)
        got_b = run(b)
    finally:
        a.close()
        b.close()
    assert np.array_equal(ia.cpu().numpy(), alone[0][0]) and np.array_equal(ib.cpu().numpy(), alone[1][0])
    assert np.array_equal(ia2.cpu().numpy(), alone[0][0]) and np.array_equal(ma2.cpu().numpy(), alone[0][1])
    for got, ref in ((got_a, alone[0]), (got_b, alone[1])):
        assert np.array_equal(got[0], ref[0]) and np.array_equal(got[1], ref[1])
        assert np.array_equal(got[2], ref[2]) and got[2].sum() > 0
        for t in range(len(ref[2])):
            k = int(ref[2][t])
            assert np.array_equal(got[3][t, :k], ref[3][t, :k]) and np.array_equal(got[4][t, :k], ref[4][t, :k])
    for t in range(len(alone[0][2])):                            # the un-synchronised interleaved decodes too
        k = int(alone[0][2][t])
        assert np.array_equal(da.cpu().numpy()[t, :k], alone[0][3][t, :k])
        k = int(alone[1][2][t])
        assert np.array_equal(db.cpu().numpy()[t, :k], alone[1][3][t, :k])


def test_stages_on_two_streams_equal_sequential():
    """one engine, the recogniser on a second stream while the table-structure net runs on the first: every stage has its
    own arena and scratch, so the results are those of the sequential calls (bit-exact)"""
    g = torch.Generator().manual_seed(9)
    gray = (torch.rand(700, L.PT_REC_H, L.PT_REC_W, generator=g) * 2 - 1).to(torch.bfloat16).cuda()
    x4 = torch.zeros(6, 512, 512, 4)
    x4[..., :3] = torch.randn(6, 512, 512, 3, generator=g) * 0.7
    x4 = x4.to(torch.bfloat16).cuda()
    e = _engine(3)
    try:
        ids0, mx0 = e.rec_forward_net(gray)
        c0, d0, l0 = e.tsr_forward_decode(x4, wiz_rev=True, vis_thresh=0.2, sync=True)
        torch.cuda.synchronize()
        s2 = torch.cuda.Stream()
        s2.wait_stream(torch.cuda.current_stream())
        for _ in range(3):
            with torch.cuda.stream(s2):
                ids1, mx1 = e.rec_forward_net(gray)
            c1, d1, l1 = e.tsr_forward_decode(x4, wiz_rev=True, vis_thresh=0.2, sync=False)
            with torch.cuda.stream(s2):
                ids2, _ = e.rec_forward_net(gray)
            torch.cuda.synchronize()
            e.check()
            assert np.array_equal(ids1.cpu().numpy(), ids0.cpu().numpy()) and np.array_equal(mx1.cpu().numpy(), mx0.cpu().numpy())
            assert np.array_equal(ids2.cpu().numpy(), ids0.cpu().numpy())
            c1h = c1.cpu().numpy()
            assert np.array_equal(c1h, c0) and c0.sum() > 0
            for t in range(len(c0)):
                k = int(c0[t])
                assert np.array_equal(d1.cpu().numpy()[t, :k], d0.cpu().numpy()[t, :k])
                assert np.array_equal(l1.cpu().numpy()[t, :k], l0.cpu().numpy()[t, :k])
    finally:
        e.close()


def test_precision_scope_serialises_other_threads_calls():
    """A stage may run in its own precision for the calls IT queues (LayoutStage.precision -> HipEngine.precision_scope).  Precision is engine-wide host
    state: while one host thread holds a scope, another thread's engine calls must wait -- never be queued in the temporary precision (ADVICE r05:
    wrong kernel namespace / (hi | lo) layout at best a format-guard error, at worst 4-channel tensors read as 8-channel).  Here a second thread flips
    the precision in a tight loop while this thread runs the recogniser: every result equals the undisturbed one bit for bit."""
    import threading
    import time
    g = torch.Generator().manual_seed(9)
    gray = (torch.rand(64, L.PT_REC_H, L.PT_REC_W, generator=g) * 2 - 1).to(torch.bfloat16).cuda()
    e = _engine(6)
    try:
        ids0, mx0 = e.rec_forward_net(gray)
        torch.cuda.synchronize()
        ids0, mx0 = ids0.cpu().numpy(), mx0.cpu().numpy()
        stop = threading.Event()
        flips = [0]

        def flipper():
            while not stop.is_set():
                with e.precision_scope(L.PT_PRECISION_BF16X3):
                    assert e.precision == L.PT_PRECISION_BF16X3
                    time.sleep(0.0005)       # another stage's calls would be queued here
                    flips[0] += 1
        th = threading.Thread(target=flipper, daemon=True)
        th.start()
        try:
            for _ in range(20):
                assert e.precision in (L.PT_PRECISION_BF16, L.PT_PRECISION_BF16X3)
                ids, mx = e.rec_forward_net(gray)      # takes the engine's lock: runs in the engine's own precision
                torch.cuda.synchronize()
                assert np.array_equal(ids.cpu().numpy(), ids0) and np.array_equal(mx.cpu().numpy(), mx0)
        finally:
            stop.set()
            th.join(timeout=10)
        assert flips[0] > 0 and e.precision == L.PT_PRECISION_BF16
        e.check()
    finally:
        e.close()
