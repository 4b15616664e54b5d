// How do the bits of a hipExtStreamCreateWithCUMask mask map onto the 8 XCDs of an MI355X?
// Launches a probe on streams whose masks enable bits [0,64), [64,256) and every 4th bit, and prints the
// number of workgroups that ran on each XCD (HW_REG_XCC_ID) and how many distinct CUs were seen.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <set>
#include <vector>
__global__ void probe(unsigned* out)
{
    // busy a little so that the workgroups spread over every enabled CU
    float x = threadIdx.x;
    for (int i = 0; i < 20000; i++) x = x * 1.0001f + 0.5f;
    if (threadIdx.x == 0) {
        const unsigned xcc = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11));      // XCC_ID[3:0]
        const unsigned hw = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11));       // HW_ID
        out[blockIdx.x] = (xcc << 24) | (hw & 0xFFFFFF) | (x == 123.f ? 1u : 0u);
    }
}
static void run(const char* name, const std::vector<uint32_t>& mask)
{
    hipStream_t s;
    if (hipExtStreamCreateWithCUMask(&s, (uint32_t)mask.size(), mask.data()) != hipSuccess) {
        printf("%s: hipExtStreamCreateWithCUMask failed\n", name);
        return;
    }
    const int nb = 4096;
    unsigned* d;
    hipMalloc(&d, nb * 4);
    probe<<<nb, 256, 0, s>>>(d);
    std::vector<unsigned> h(nb);
    hipStreamSynchronize(s);
    hipMemcpy(h.data(), d, nb * 4, hipMemcpyDeviceToHost);
    int per[16] = {0};
    std::set<unsigned> cus;
    for (unsigned v : h) {
        per[(v >> 24) & 15]++;
        // HW_ID: cu_id [11:8], sh_id [12], se_id [15:13]
        cus.insert(((v >> 24) << 16) | ((v >> 8) & 0xFF));
    }
    printf("%-14s XCD workgroups:", name);
    for (int i = 0; i < 8; i++) printf(" %4d", per[i]);
    printf("   distinct (xcd,se,sh,cu): %zu\n", cus.size());
    hipFree(d);
    hipStreamDestroy(s);
}
int main()
{
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    printf("CUs: %d\n", p.multiProcessorCount);
    std::vector<uint32_t> all(8, 0xFFFFFFFFu), lo(8, 0), hi(8, 0), every4(8, 0);
    for (int b = 0; b < 256; b++) {
        if (b < 64) lo[b / 32] |= 1u << (b % 32); else hi[b / 32] |= 1u << (b % 32);
        if (b % 4 == 0) every4[b / 32] |= 1u << (b % 32);
    }
    run("all", all);
    run("bits 0..63", lo);
    run("bits 64..255", hi);
    run("every 4th bit", every4);
    return 0;
}
