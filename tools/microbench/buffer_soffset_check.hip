// Does the raw-buffer range check of gfx950 cover the scalar offset?  A 4 KiB allocation filled with its own dword indices, a descriptor
// that claims 1 KiB of it, and loads whose vector / scalar offsets land inside and outside the 1 KiB: a checked access returns 0.
//   hipcc --offload-arch=gfx950 -O2 tools/microbench/buffer_soffset_check.hip -o tools/microbench/buffer_soffset_check && ./buffer_soffset_check
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4_t;
__global__ void probe(const uint32_t* buf, uint32_t* out, char* lds_out) {
  __shared__ __attribute__((aligned(16))) char lds[1024];
  const auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)buf, 0, 1024u, 0x00020000);
  const int lane = threadIdx.x;
  if (lane == 0) {
    out[0] = __builtin_amdgcn_raw_buffer_load_b32(rs, 512, 0, 0);      // voffset inside, soffset 0: dword 128
    out[1] = __builtin_amdgcn_raw_buffer_load_b32(rs, 2048, 0, 0);     // voffset outside: 0
    out[2] = __builtin_amdgcn_raw_buffer_load_b32(rs, 512, 2048, 0);   // voffset inside, voffset + soffset outside: 0 if soffset is checked, 640 if not
    out[3] = __builtin_amdgcn_raw_buffer_load_b32(rs, 0, 2048, 0);     // soffset alone outside
    out[4] = __builtin_amdgcn_raw_buffer_load_b32(rs, 1020, 8, 0);     // last dword inside + soffset: 257 if unchecked
  }
  // the same through the LDS-DMA form
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) char*)lds, 16, (unsigned)(lane * 16) % 1024u, 2048, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (lane == 0) out[5] = *(const uint32_t*)(lds);   // dword 512 if unchecked, 0 if checked
}
int main() {
  uint32_t *d, *o, h[1024], r[8] = {};
  for (int i = 0; i < 1024; ++i) h[i] = i;
  hipMalloc(&d, 4096); hipMalloc(&o, 64);
  hipMemcpy(d, h, 4096, hipMemcpyHostToDevice);
  hipMemset(o, 0xFF, 64);
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, o, nullptr);
  hipMemcpy(r, o, 32, hipMemcpyDeviceToHost);
  printf("inside %u | voffset outside %u | voffset in + soffset out %u | soffset out %u | edge %u | lds-dma soffset out %u\n", r[0], r[1], r[2], r[3], r[4], r[5]);
  printf("=> the range check %s the scalar offset\n", (r[2] == 0 && r[3] == 0) ? "COVERS" : "does NOT cover");
  return 0;
}
