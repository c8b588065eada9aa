// TEST INFRASTRUCTURE ONLY: exported switches of the host emulator (tests/hipemu/hip/hip_runtime.h).
#include <hip/hip_runtime.h>

extern "C" {
// 1: launches of 2..256 work-groups run every work-group on its own OS thread (needed by kernels whose work-groups wait
// for each other); 0: work-groups run one after another on the caller's thread (default, deterministic, cheap).
void hipemu_set_concurrent(int on) { hipemu::concurrent_flag() = on; }
int hipemu_get_concurrent(void) { return hipemu::concurrent_flag(); }
}
