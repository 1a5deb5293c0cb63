#include "wgrad_impl.h"
namespace mi {
int launch_wgrad_f32(const WgradArgs& a, hipStream_t s, const char** name) { return launch_wgrad_t<float>(a, s, name); }
}
