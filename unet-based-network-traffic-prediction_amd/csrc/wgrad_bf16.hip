#include "wgrad_impl.h"
namespace mi {
int launch_wgrad_bf16(const WgradArgs& a, hipStream_t s, const char** name) { return launch_wgrad_t<__bf16>(a, s, name); }
}
