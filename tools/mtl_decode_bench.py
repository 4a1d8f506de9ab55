"""tools/mtl_decode_bench.py [n_tables] [max_len]: wall time of the MtlTabNet decoders (pt_tsr_mtl_structure + pt_tsr_mtl_cells) on random features,
seeded weights at the real vocabulary sizes -- run under `rocprofv3 --kernel-trace --stats` to compare the kernels' own time with the wall time
(launch-bound or not)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pdf_table_amd import lib as L                                   # noqa: E402
from pdf_table_amd.engine import HipEngine                           # noqa: E402
from pdf_table_amd.mtl_stage import MtlTabNetConvertor               # noqa: E402
from pdf_table_amd.synth_weights import mtl_tabnet_decoder_state_dict  # noqa: E402
from pdf_table_amd.weights import pack_mtl_decoder                   # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 87
max_len = int(sys.argv[2]) if len(sys.argv) > 2 else 500
conv = MtlTabNetConvertor(max_seq_len=max_len)
eng = HipEngine(0)
sd = mtl_tabnet_decoder_state_dict(seed=43, num_classes=conv.num_classes(), num_classes_cell=conv.num_classes_cell())
if "--no-pad" in sys.argv:      # a trained model never emits <PAD>: keep the loops in their KV-cached mode (one query per table and step)
    cfg = conv.decoder_cfg()
    for key, pad in (("cls_fc.bias", cfg["pad"]), ("cell_fc.bias", cfg["pad_cell"])):
        sd[key] = sd[key].clone()
        sd[key][pad] = -1e4
eng.load_weights(L.PT_MODEL_MTL_DECODER, pack_mtl_decoder(sd, conv.decoder_cfg()))
f3 = torch.randn(n, 3600, 512, generator=torch.Generator().manual_seed(0)).cuda()
for prec, name in ((L.PT_PRECISION_BF16, "bf16"), (L.PT_PRECISION_BF16, "bf16 + fp8 keys / values"), (L.PT_PRECISION_BF16X3, "bf16x3")):
    eng.set_precision(prec)
    eng.set_mtl_kv_fp8("fp8" in name)
    eng.mtl_decode(f3)
    torch.cuda.synchronize()
    if "--prof" in sys.argv:          # PT_PROF_VERBOSE=1: HIP-event time per launch label (serialises the launches: not a throughput figure)
        eng.profile_enable(True)
        eng.mtl_decode(f3)
        print(f"-- per-label times, {name} --", file=sys.stderr, flush=True)
        eng.profile_read()
        eng.profile_enable(False)
    t0 = time.perf_counter()
    out = eng.mtl_decode(f3)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    steps = int(out["lens"].max())
    print(f"mtl decode {name}: {n} tables, {steps} structure positions, {int(out['cell_counts'].sum())} cells: {dt * 1e3:.1f} ms = {dt / steps * 1e6:.0f} us per step, "
          f"{n / dt:.0f} tables/s")
