#include <hip/hip_runtime.h>
__global__ void k(const float4* a, float4* b, float4* c, const float4* d) {
    int i = threadIdx.x + blockIdx.x * 256;
    float4 x = a[i];
    float4 y = d[i];
    b[i] = make_float4(1.f, 2.f, 3.f, 4.f);
    b[i + 4096] = make_float4(1.f, 2.f, 3.f, 5.f);
    b[i + 8192] = y;
    c[i] = make_float4(x.x + 1.f, x.y, x.z, x.w);
}
