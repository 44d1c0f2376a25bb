// Where do the blocks of a grid land?  Block b records (XCC id, SE id, CU id) -- the question behind static row-tile
// schedules that want "one big and one small tile per CU" (chain.hip).  hipcc --offload-arch=gfx950 -O2 -o census_lab census_lab.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <map>
__global__ void census(unsigned* out, int spin) {
  unsigned hwid, xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  if (threadIdx.x == 0) { out[2 * blockIdx.x] = hwid; out[2 * blockIdx.x + 1] = xcc; }
  // stay resident so that the whole grid co-exists
  long long t0 = wall_clock64();
  while (wall_clock64() - t0 < spin) {}
}
int main(int argc, char** argv) {
  int threads = argc > 1 ? atoi(argv[1]) : 512, lds = argc > 2 ? atoi(argv[2]) : 66 * 1024;
  for (int grid : {256, 448, 512, 600, 799}) {
    unsigned* d;
    hipMalloc(&d, grid * 8);
    hipFuncSetAttribute((const void*)census, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipLaunchKernelGGL(census, dim3(grid), dim3(threads), lds, 0, d, 2000000);
    std::vector<unsigned> h(2 * grid);
    hipMemcpy(h.data(), d, grid * 8, hipMemcpyDeviceToHost);
    std::map<unsigned, std::vector<int>> cu;      // (xcc, se, cu) -> blocks
    for (int b = 0; b < grid; ++b) {
      unsigned hw = h[2 * b], x = h[2 * b + 1] & 0xf;
      unsigned cuid = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 0x7;
      cu[(x << 12) | (se << 8) | (sh << 4) | cuid].push_back(b);
    }
    std::map<size_t, int> hist;
    for (auto& kv : cu) hist[kv.second.size()]++;
    printf("grid %d threads %d lds %d: distinct CUs %zu; blocks per CU histogram:", grid, threads, lds, cu.size());
    for (auto& kv : hist) printf(" %zu:%d", kv.first, kv.second);
    printf("\n  first CUs:");
    int n = 0;
    for (auto& kv : cu) { if (n++ >= 6) break; printf(" [%x:", kv.first); for (int b : kv.second) printf(" %d", b); printf("]"); }
    printf("\n");
    hipFree(d);
  }
  return 0;
}
