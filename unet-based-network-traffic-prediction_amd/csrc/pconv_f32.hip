#include "pconv_impl.h"
namespace mi {
int launch_pconv_f32(const PconvArgs& a, hipStream_t s, const char** name) { return launch_pconv_t<float>(a, s, name); }
}
