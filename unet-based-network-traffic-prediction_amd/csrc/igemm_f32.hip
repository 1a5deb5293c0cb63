#include "igemm_impl.h"
namespace mi {
int launch_igemm_f32(const IgemmArgs& a, hipStream_t s, const char** name) { return launch_igemm_t<float>(a, s, name); }
}
