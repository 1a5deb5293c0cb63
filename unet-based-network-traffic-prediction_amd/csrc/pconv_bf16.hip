#include "pconv_impl.h"
namespace mi {
int launch_pconv_bf16(const PconvArgs& a, hipStream_t s, const char** name) { return launch_pconv_t<__bf16>(a, s, name); }
}
