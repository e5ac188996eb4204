#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(256) void k(int *o) { extern __shared__ int s[]; s[threadIdx.x] = 1; __syncthreads(); o[threadIdx.x] = s[255 - threadIdx.x]; }
int main() {
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  printf("sharedMemPerMultiprocessor %zu  maxSharedMemoryPerMultiProcessor %zu sharedMemPerBlock %zu\n", (size_t)p.sharedMemPerMultiprocessor, (size_t)p.maxSharedMemoryPerMultiProcessor, (size_t)p.sharedMemPerBlock);
  for (size_t sm : {32768ul, 39936ul, 40448ul, 40960ul, 41472ul, 50176ul, 53248ul}) { int n = 0; hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k, 256, sm); printf("dyn LDS %zu -> %d blocks/CU\n", sm, n); }
}
