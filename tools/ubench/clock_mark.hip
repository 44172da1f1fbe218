// clock_mark.hip -- diagnostic only (tools/exp_group_timeline.py): a one-thread kernel that writes the 100 MHz wall clock into
// a slot of a device buffer; launched (and captured into the step graph) in front of / behind the launches whose position
// in time is asked for.   hipcc --offload-arch=gfx950 -O2 -fPIC -shared -o tools/ubench/libclock_mark.so tools/ubench/clock_mark.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
__global__ void k_clock_mark(unsigned long long* buf, int slot) {
  if (threadIdx.x == 0 && blockIdx.x == 0) buf[slot] = wall_clock64();
}
extern "C" int clock_mark(unsigned long long* buf, int slot, void* stream) {
  hipLaunchKernelGGL(k_clock_mark, dim3(1), dim3(64), 0, (hipStream_t)stream, buf, slot);
  return (int)hipGetLastError();
}
