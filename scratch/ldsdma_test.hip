// micro-test: does buffer_load ... lds with an out-of-range offset write ZEROS into LDS?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
__global__ void k(const uint32_t* src, uint32_t nbytes, uint32_t* out) {
    __shared__ __attribute__((aligned(16))) uint32_t lds[64 * 4 * 2];
    for (int i = threadIdx.x; i < 512; i += 64) lds[i] = 0xDEADBEEF;
    __syncthreads();
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, nbytes, 0x00020000);
    const int lane = threadIdx.x;
    // even lanes in range (permuted source: lane reads chunk (lane ^ 3)), odd lanes out of range
    uint32_t off = (lane & 1) ? 0x7FFFFFF0u : (uint32_t)((lane ^ 3) * 16);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)lds, 16, off, 0, 0, 0);
    // second instruction to the second KB, all lanes in range
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(lds + 256), 16, lane * 16, 0, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 512; i += 64) out[i] = lds[i];
}
int main() {
    std::vector<uint32_t> h(4096);
    for (int i = 0; i < 4096; i++) h[i] = i + 1;
    uint32_t *d, *o;
    hipMalloc(&d, 4096 * 4); hipMalloc(&o, 512 * 4);
    hipMemcpy(d, h.data(), 4096 * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, 4096 * 4, o);
    std::vector<uint32_t> r(512);
    hipMemcpy(r.data(), o, 512 * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int lane = 0; lane < 64; lane++)
        for (int j = 0; j < 4; j++) {
            uint32_t want = (lane & 1) ? 0u : (uint32_t)((lane ^ 3) * 4 + j + 1);
            if (r[lane * 4 + j] != want) { if (bad < 8) printf("lane %d j %d got %08x want %08x\n", lane, j, r[lane * 4 + j], want); bad++; }
            uint32_t want2 = lane * 4 + j + 1;
            if (r[256 + lane * 4 + j] != want2) { if (bad < 8) printf("B lane %d j %d got %08x want %08x\n", lane, j, r[256 + lane * 4 + j], want2); bad++; }
        }
    printf("ldsdma OOB->zero test: %s (%d mismatches)\n", bad ? "FAIL" : "PASS", bad);
    return bad != 0;
}
