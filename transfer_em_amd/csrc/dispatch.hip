// dispatch.hip -- public convolution entry points: pick the LDS/MFMA-tiled kernel when the
// geometry is one it was built for, otherwise the shape-generic direct kernel.
#include "tem_common.h"

int tem_conv_lds_try(const tem_conv_args *a, hipStream_t st, bool dry);      // conv_lds.hip
int tem_conv_c1_try(const tem_conv_args *a, hipStream_t st, bool dry);       // stencil_c1.hip
int tem_convT_mfma_try(const tem_conv_args *a, hipStream_t st, bool dry);    // convT_mfma.hip
int tem_conv_wino_try(const tem_conv_args *a, hipStream_t st, bool dry);     // wino.hip
int tem_conv_s2_try(const tem_conv_args *a, hipStream_t st, bool dry);       // conv_s2.hip
int tem_conv_c1out_try(const tem_conv_args *a, hipStream_t st, bool dry);    // c1out_mfma.hip

extern "C" int tem_conv(const tem_conv_args *a, tem_stream_t stream) {
  TEM_CLEAR_ERR();
  if (a && tem_view_ok(a->in0) && tem_view_ok(a->out0) && a->w) {
    if (a->w_layout == TEM_W_WINOGRAD) return tem_conv_wino_try(a, (hipStream_t)stream, false);   // no other reader of that layout
    int rc = tem_conv_s2_try(a, (hipStream_t)stream, false);
    if (rc != TEM_EUNSUPPORTED) return rc;
    rc = tem_conv_lds_try(a, (hipStream_t)stream, false);
    if (rc != TEM_EUNSUPPORTED) return rc;
    rc = tem_conv_c1out_try(a, (hipStream_t)stream, false);
    if (rc != TEM_EUNSUPPORTED) return rc;
    rc = tem_conv_c1_try(a, (hipStream_t)stream, false);
    if (rc != TEM_EUNSUPPORTED) return rc;
  }
  return tem_conv_direct(a, stream);
}

extern "C" int tem_conv_transpose(const tem_conv_args *a, tem_stream_t stream) {
  TEM_CLEAR_ERR();
  if (a && tem_view_ok(a->in0) && tem_view_ok(a->out0) && a->w) {
    int rc = tem_convT_mfma_try(a, (hipStream_t)stream, false);
    if (rc != TEM_EUNSUPPORTED) return rc;
  }
  return tem_conv_transpose_direct(a, stream);
}

int tem_conv_lds_describe(const tem_conv_args *a, char *buf, int len);      // conv_lds.hip
int tem_conv_c1_describe(const tem_conv_args *a, char *buf, int len);       // stencil_c1.hip
int tem_convT_mfma_describe(const tem_conv_args *a, char *buf, int len);    // convT_mfma.hip
int tem_conv_wino_describe(const tem_conv_args *a, char *buf, int len);     // wino.hip
int tem_conv_s2_describe(const tem_conv_args *a, char *buf, int len);       // conv_s2.hip
int tem_conv_c1out_describe(const tem_conv_args *a, char *buf, int len);    // c1out_mfma.hip
int tem_bww_lds_describe(const tem_bww_args *a, char *buf, int len);        // bww_lds.hip
int tem_bww_c1_describe(const tem_bww_args *a, char *buf, int len);         // bww_c1.hip
int tem_bww_s2_describe(const tem_bww_args *a, char *buf, int len);         // bww_s2.hip
int tem_conv_direct_describe(const tem_conv_args *a, char *name, int len);  // conv_direct.hip

extern "C" int tem_conv_is_tiled(const tem_conv_args *a, int32_t transposed, char *name, int32_t name_len) {
  if (!a || !tem_view_ok(a->in0) || !tem_view_ok(a->out0) || !a->w) return TEM_EINVAL;
  if (name && name_len > 0) name[0] = 0;
  if (transposed) return tem_convT_mfma_describe(a, name, name_len) == TEM_OK ? 1 : 0;
  if (a->w_layout == TEM_W_WINOGRAD) return tem_conv_wino_describe(a, name, name_len) == TEM_OK ? 1 : 0;
  if (tem_conv_s2_describe(a, name, name_len) == TEM_OK) return 1;
  if (tem_conv_lds_describe(a, name, name_len) == TEM_OK) return 1;
  if (tem_conv_c1out_describe(a, name, name_len) == TEM_OK) return 1;
  if (tem_conv_c1_describe(a, name, name_len) == TEM_OK) return 1;
  if (name && name_len > 0) tem_conv_direct_describe(a, name, name_len);   // name of the direct kernel that will run
  return 0;
}

extern "C" int tem_bww_is_tiled(const tem_bww_args *a, char *name, int32_t name_len) {
  if (!a || !tem_view_ok(a->in0) || !tem_view_ok(a->dout)) return TEM_EINVAL;
  if (name && name_len > 0) name[0] = 0;
  if (tem_bww_c1_describe(a, name, name_len) == TEM_OK) return 1;
  if (tem_bww_s2_describe(a, name, name_len) == TEM_OK) return 1;
  return tem_bww_lds_describe(a, name, name_len) == TEM_OK ? 1 : 0;
}
