// api_common.cpp -- the format-independent corner of the C ABI (compiled once; c_api.hip and graph_ops.hip are compiled per activation format and
// reached through api_dispatch.cpp): the thread-local error string every layer of the library writes, and the ABI version.
#include <stdarg.h>
#include <stdio.h>

#include "../../include/pdftable_hip.h"

static thread_local char g_err[1024] = "";

void pt_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" {

const char* pt_last_error(void) { return g_err; }

// 14: PT_PRECISION_F16 (every kernel instantiated for IEEE-half storage; blobs carry their format); 13: pt_engine_set_dcn_mfma; 12: pt_op_dcn (the fused
// modulated deformable convolution as a single operator); 11: pt_engine_set_mtl_kv_fp8; 10: pt_tsr_mtl_preprocess / decoder_config / structure / cells
// (MtlTabNet decoders); 9: pt_rec_cvit_* (ConvNextViT recogniser); 8: pt_op_dwconv / add / maxpool / avgpool / chan_mean / scale_channels / act (generic
// ONNX executor); 7: pt_rec_pp_preprocess*; 6: pt_engine_set_lstm_cluster, pt_engine_check clears what it reports
int pt_abi_version(void) { return 14; }

}  // extern "C"
