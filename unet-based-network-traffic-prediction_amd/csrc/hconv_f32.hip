#include "hconv_impl.h"
namespace mi {
int launch_hconv_f32(const HconvArgs& a, hipStream_t s, const char** name) { return launch_hconv_t<float>(a, s, name); }
}
