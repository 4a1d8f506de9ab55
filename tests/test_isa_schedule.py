"""The schedules the hot kernels rely on, read from the compiler's own output (no GPU): software-pipelined LDS reads stay pipelined (an MFMA does not wait
`lgkmcnt(0)` on a read issued just in front of it) and no FLAT-encoded memory instruction sneaks into them (with one in flight hipcc's wait-count pass orders
every LDS dependency; round 6 found three kernels that way).  tools/isa_wait_audit.py is the tool; this pins its readings of the kernels that were fixed."""
import importlib.util
import os
import shutil

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _tool():
    spec = importlib.util.spec_from_file_location("isa_wait_audit", os.path.join(ROOT, "tools", "isa_wait_audit.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


@pytest.mark.skipif(not os.path.exists("/opt/rocm/bin/hipcc") or shutil.which("c++filt") is None, reason="needs hipcc and c++filt")
def test_pinned_kernels_keep_their_lds_reads_in_flight():
    t = _tool()
    csrc = os.path.join(ROOT, "pdf_table_amd", "csrc")
    res = t.audit([os.path.join(csrc, f) for f in ("rec_kernels.hip", "lore_kernels.hip", "det_kernels.hip", "conv_igemm.hip")])

    def pick(sub):
        hits = {k: v for k, v in res.items() if sub in k}
        assert hits, sub
        return hits

    # (kernel substring, largest share of MFMAs that may wait lgkmcnt(0) on a just-issued read)
    for sub, bound in (("cls_argmax_dma_kernel<0>", 0.10), ("cls_argmax_dma_kernel<1>", 0.10), ("lstm_cluster_kernel<3>", 0.10), ("lstm_cluster_kernel<2>", 0.10),
                       ("dla_thin_chain16_kernel", 0.30), ("crnn_conv01_kernel", 0.10), ("conv3x3_pipe_persist_kernel<2, 8, 1>", 0.10),
                       ("conv3x3_pipe_kernel<1, 4, true, 0, true, 1, false>", 0.10), ("conv3x3_ws64_kernel", 0.10)):
        for k, (mf, w, fl) in pick(sub).items():
            assert mf >= 16 and w / mf <= bound, (k, mf, w)
    # no FLAT-encoded loads / stores anywhere in these translation units (LDS and global pointers keep their address spaces)
    flat = {k: v[2] for k, v in res.items() if v[2]}
    assert not flat, flat
