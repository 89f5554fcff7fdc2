// Probe: operand layout of v_smfmac_f32_32x32x32_bf16 (gfx950), found by experiment -- which (lane half, element) of the
// dense B operand meets which (lane half, compressed slot, 2-bit index) of the sparse A operand.  One wave; B is one-hot
// in (half hb, element e) for every column, A is one-hot in (half ha, slot s) with index value iv for every row; a
// non-zero D means the two address the same k.  Prints, per (hb, e), the matching (ha, s, iv).  Diagnostic only.
//   hipcc --offload-arch=gfx950 -O2 scripts/probes/smfmac_layout.hip -o scripts/probes/_bin/smfmac_layout
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8;
typedef __attribute__((__vector_size__(16 * sizeof(__bf16)))) __bf16 bf16x16;
typedef __attribute__((__vector_size__(16 * sizeof(float)))) float f32x16;

__global__ void k(int *out)
{
    const int lane = threadIdx.x, half = lane >> 5;
    for (int hb = 0; hb < 2; ++hb)
        for (int e = 0; e < 16; ++e) {
            bf16x16 b;
            for (int i = 0; i < 16; ++i) b[i] = (half == hb && i == e) ? (__bf16)1.0f : (__bf16)0.0f;
            for (int ha = 0; ha < 2; ++ha)
                for (int s = 0; s < 8; ++s)
                    for (int iv = 0; iv < 4; ++iv) {
                        bf16x8 a;
                        for (int i = 0; i < 8; ++i) a[i] = (half == ha && i == s) ? (__bf16)1.0f : (__bf16)0.0f;
                        // every slot's index: slot s gets iv, its partner in the group another value, the rest (0, 1)
                        int idx = 0;
                        for (int i = 0; i < 8; ++i) {
                            int v = (i & 1) ? 1 : 0;
                            if (i == s) v = iv;
                            else if (i == (s ^ 1)) v = (iv + 1 + (i & 1)) & 3, v = v == iv ? (v + 1) & 3 : v;
                            idx |= v << (2 * i);
                        }
                        f32x16 c;
                        for (int i = 0; i < 16; ++i) c[i] = 0.f;
                        c = __builtin_amdgcn_smfmac_f32_32x32x32_bf16(a, b, c, idx, 0, 0);
                        float sum = 0.f;
                        for (int i = 0; i < 16; ++i) sum += c[i];
                        // every lane of a wave sees 16 of the 32 rows of its column: all rows are alike here
                        if (lane == 0) out[((hb * 16 + e) * 2 + ha) * 32 + s * 4 + iv] = sum != 0.f;
                    }
        }
}

int main()
{
    int *d;
    if (hipMalloc(&d, 32 * 2 * 32 * 4) != hipSuccess) return 1;
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    static int h[32 * 2 * 32];
    if (hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess) return 1;
    for (int hb = 0; hb < 2; ++hb)
        for (int e = 0; e < 16; ++e) {
            printf("B half %d element %2d:", hb, e);
            for (int ha = 0; ha < 2; ++ha)
                for (int s = 0; s < 8; ++s)
                    for (int iv = 0; iv < 4; ++iv)
                        if (h[((hb * 16 + e) * 2 + ha) * 32 + s * 4 + iv]) printf("  A half %d slot %d idx %d", ha, s, iv);
            printf("\n");
        }
    return 0;
}
