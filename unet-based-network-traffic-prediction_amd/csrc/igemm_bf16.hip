#include "igemm_impl.h"
namespace mi {
int launch_igemm_bf16(const IgemmArgs& a, hipStream_t s, const char** name) { return launch_igemm_t<__bf16>(a, s, name); }
}
