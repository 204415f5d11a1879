// LDS-DMA (buffer_load ... lds) throughput probe: what one CU sustains from L2-resident operand panels, by access pattern.
//   pattern 0: a 1-KiB piece = 16 rows x 64 B (the BK = 32 K tile of the GEMM kernels: half a 128-B line per row)
//   pattern 1: a 1-KiB piece =  8 rows x 128 B (BK = 64: whole lines)
//   pattern 2: a 1-KiB piece =  4 rows x 256 B, pattern 3: 1 row x 1 KiB (fully contiguous)
// Every workgroup (NW waves, one or two per CU) walks ROWS rows x KB bytes of its own panel (L2-resident after the first sweep)
// for `sweeps` sweeps; pieces are issued round-robin by its waves with at most `inflight` outstanding per wave.
// Build: hipcc --offload-arch=gfx950 -O3 tools/dma_probe.hip -o tools/bin/dma_probe ; run: tools/bin/dma_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define LDS_AS __attribute__((address_space(3)))

template <int PAT, int INFLIGHT>
__global__ void __launch_bounds__(512) dma_kernel(const char* base, long long panel_bytes, int rows, int row_bytes, int sweeps, int shared_panel, unsigned* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nw = blockDim.x >> 6;
    const char* p = base + (shared_panel ? 0 : (long long)blockIdx.x * panel_bytes);
    const unsigned long long v = (unsigned long long)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    const auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)(((unsigned long long)hi << 32) | lo), 0, 0x80000000, 0x00020000);
    constexpr int RPP = PAT == 0 ? 16 : PAT == 1 ? 8 : PAT == 2 ? 4 : 1;      // rows per piece
    constexpr int BPR = 1024 / RPP;                                           // bytes per row and piece
    const int r_in = lane / (64 / RPP), c_in = (lane % (64 / RPP)) * 16;
    const int row_pieces = rows / RPP, col_steps = row_bytes / BPR;
    char* dst = smem + wave * 8192;
    int slot = 0;
    for (int s = 0; s < sweeps; ++s) {
        for (int cs = 0; cs < col_steps; ++cs) {                              // walk along K like a GEMM: all rows of one K tile, then the next
            for (int rp = wave; rp < row_pieces; rp += nw) {
                const unsigned off = (unsigned)((rp * RPP + r_in) * row_bytes + cs * BPR + c_in);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (LDS_AS void*)(dst + slot * 1024), 16, (int)off, 0, 0, 0);
                slot = (slot + 1) & 7;
                if (INFLIGHT == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                if (INFLIGHT == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                if (INFLIGHT == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) sink[blockIdx.x] = *(unsigned*)smem;
}

template <int PAT, int INF>
float run(const char* buf, long long panel_bytes, int rows, int row_bytes, int sweeps, int shared_panel, int threads, int blocks, size_t lds, unsigned* sink) {
    hipFuncSetAttribute((const void*)dma_kernel<PAT, INF>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL((dma_kernel<PAT, INF>), dim3(blocks), dim3(threads), lds, 0, buf, panel_bytes, rows, row_bytes, 2, shared_panel, sink);   // warm (L2 fill)
    hipEventRecord(a);
    hipLaunchKernelGGL((dma_kernel<PAT, INF>), dim3(blocks), dim3(threads), lds, 0, buf, panel_bytes, rows, row_bytes, sweeps, shared_panel, sink);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms = 0; hipEventElapsedTime(&ms, a, b);
    return ms;
}

int main() {
    const int rows = 576, row_bytes = 1536;                  // (320 + 256) rows x K = 768 bf16: the operand panels of one 320 x 256 tile
    const long long panel = (long long)rows * row_bytes;     // 864 KiB per workgroup
    const int cus = 256;
    char* buf; unsigned* sink;
    hipMalloc(&buf, panel * cus * 2); hipMemset(buf, 1, panel * cus * 2); hipMalloc(&sink, 4096 * 4);
    const int sweeps = 40;
    printf("pattern rows/piece  wgs/CU thr inflight shared   ms    GB/s/CU  B/clk/CU@2.1GHz  cyc/piece/CU\n");
    for (int cfg = 0; cfg < 2; ++cfg) {                       // 0: one 512-thread WG per CU (128 KiB LDS), 1: two 256-thread WGs per CU (72 KiB each)
        const int threads = cfg ? 256 : 512, blocks = cfg ? 2 * cus : cus;
        const size_t lds = cfg ? 72 * 1024 : 128 * 1024;
        for (int sh = 0; sh < 2; ++sh)
        for (int pat = 0; pat < 4; ++pat)
        for (int inf : {4, 16}) {
            float ms = 0;
#define RUN(P, I) if (pat == P && inf == I) ms = run<P, I>(buf, panel, rows, row_bytes, sweeps, sh, threads, blocks, lds, sink);
            RUN(0, 4) RUN(0, 16) RUN(1, 4) RUN(1, 16) RUN(2, 4) RUN(2, 16) RUN(3, 4) RUN(3, 16)
            const double bytes_cu = (double)panel * sweeps * (cfg ? 2 : 1);
            const double gbs = bytes_cu / (ms * 1e-3) / 1e9;
            printf("%7d %10d %7d %4d %8d %6d %7.3f %8.1f %10.1f %14.1f\n", pat, pat == 0 ? 16 : pat == 1 ? 8 : pat == 2 ? 4 : 1, cfg + 1, threads, inf, sh, ms, gbs, gbs / 2.1,
                   1024.0 / (gbs / 2.1));
        }
    }
    return 0;
}
