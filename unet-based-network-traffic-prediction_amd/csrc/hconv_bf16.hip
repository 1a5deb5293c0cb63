#include "hconv_impl.h"
namespace mi {
int launch_hconv_bf16(const HconvArgs& a, hipStream_t s, const char** name) { return launch_hconv_t<__bf16>(a, s, name); }
}
