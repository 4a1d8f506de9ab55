"""Architecture table of ``DBNasModel`` (db_net/dbnet.py:693-712): the searched ProxylessNAS backbone
``CompactDetBackbone(width_stages=[32, 64, 96, 128], input_channel=32)`` (db_net/proxyless.py:92-178) and the
``LightSegDetector`` decoder (dbnet.py:338-481).  Shared by the synthetic checkpoint, the weight packer and the
CPU oracle; the HIP graph (csrc/dbnas_model.hip) reads the same table from the blob's ``arch`` tensor.

The backbone is a list of 24 blocks: per stage five inverted-residual conv blocks and one squeeze-excite block.
Which operator each block uses is the search result hard-coded in proxyless.py:118-127 (``conv_op_ids`` indexing
``conv_candidates`` / ``se_candidates``):

* ``rep`` -- ``MBInvertedRepConvLayer`` (db_net/layers.py:669-745): 1x1 expand + BN + PReLU, a SUM of depthwise
  convs of several kernel sizes (each with its own BN), PReLU, 1x1 project + BN;
* ``mb``  -- ``MBInvertedConvLayer`` (layers.py:93-160): 1x1 expand + BN + PReLU, depthwise k x k + BN + PReLU,
  1x1 project + BN;
* ``se``  -- ``SELayer`` (layers.py:469-490) inside a residual block: x + x * sigmoid(fc2(relu(fc1(mean(x))))).

A conv block has the identity shortcut when stride is 1 and cin == cout (proxyless.py:150-153)."""
from __future__ import annotations

from typing import Dict, List

WIDTH_STAGES = (32, 64, 96, 128)
INPUT_CHANNEL = 32
INNER_CHANNELS = 64          # LightSegDetector(inner_channels=64, dw_kernel_size=5), dbnet.py:703-710
DW_KERNEL = 5

# conv_candidates index -> (kind, depthwise kernel sizes, expand ratio)   (proxyless.py:107-114 + mix_ops.py:36-230)
_CONV_OPS = {0: ("mb", (5,), 2), 1: ("mb", (5,), 4), 2: ("mb", (3,), 2), 3: ("mb", (3,), 4),
             15: ("rep", (3, 5), 2), 16: ("rep", (3, 5), 4), 17: ("rep", (1, 3, 5), 2), 18: ("rep", (1, 3, 5), 4)}
_SE_OPS = {0: 2, 1: 4, 2: 8}   # se_candidates index -> squeeze factor (proxyless.py:116)
CONV_OP_IDS = (15, 17, 17, 17, 17, 0, 16, 16, 18, 18, 16, 2, 16, 18, 16, 18, 18, 2, 1, 18, 18, 18, 16, 2)
N_CELL = 5


def dbnas_blocks() -> List[Dict]:
    blocks: List[Dict] = []
    cin = INPUT_CHANNEL
    for width in WIDTH_STAGES:
        for i in range(N_CELL):
            kind, sizes, expand = _CONV_OPS[CONV_OP_IDS[len(blocks)]]
            stride = 2 if i == 0 else 1
            blocks.append({"kind": kind, "cin": cin, "cout": width, "mid": cin * expand, "sizes": sizes, "stride": stride,
                           "shortcut": stride == 1 and cin == width})
            cin = width
        blocks.append({"kind": "se", "cin": cin, "cout": cin, "squeeze": cin // _SE_OPS[CONV_OP_IDS[len(blocks)]]})
    return blocks


OUTPUT_BLOCKS = (5, 11, 17, 23)     # NasRecBackbone.forward: every len(blocks)/4-th block (proxyless.py:21-31)
