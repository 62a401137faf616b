// developer tool: what a raw buffer access does when its per-lane offset is outside the descriptor's num_records --
// the column kernels park the stores of lanes that own no output cell there (pomgpu_internal.hpp: BOFF_NONE) instead
// of branching around them.  Expected on gfx950: store discarded, load returns 0.
//   hipcc --offload-arch=gfx950 -O3 -o buf_oob buf_oob.hip && ./buf_oob
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
__global__ void k(double *p, unsigned n_in, double *out, unsigned soff) {
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void *)p, 0, n_in * 8, 0x00020000);
  const unsigned lane = threadIdx.x;
  // lanes 0..31: in range; 32..47: first bytes past the end; 48..63: 0xFFFFFFF0
  const unsigned voff = lane < 32 ? lane * 8 : (lane < 48 ? (n_in + lane - 32) * 8 : 0xFFFFFFF0u);
  u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(r, (int)voff, (int)soff, 0);
  out[lane] = __hiloint2double((int)v.y, (int)v.x);
  u32x2 w; w.x = 0x12345678u; w.y = 0x40000000u + lane;
  __builtin_amdgcn_raw_buffer_store_b64(w, r, (int)voff, (int)soff, 0);
}
int main() {
  const unsigned n_in = 64, n_all = 256;
  double *p, *out, h[n_all], o[64];
  hipMalloc(&p, n_all * 8); hipMalloc(&out, 64 * 8);
  for (unsigned soff : {0u, 128u}) {
    for (unsigned n = 0; n < n_all; n++) h[n] = 1000.0 + n;
    hipMemcpy(p, h, sizeof h, hipMemcpyHostToDevice);
    k<<<1, 64>>>(p, n_in, out, soff);
    hipMemcpy(h, p, sizeof h, hipMemcpyDeviceToHost); hipMemcpy(o, out, sizeof o, hipMemcpyDeviceToHost);
    int changed_in = 0, changed_out = 0, zero_loads = 0, good_loads = 0;
    for (unsigned n = 0; n < n_all; n++) if (h[n] != 1000.0 + n) { if (n < n_in) changed_in++; else changed_out++; }
    for (int l = 0; l < 64; l++) { if (l < 32 && o[l] == 1000.0 + l + soff / 8) good_loads++; if (l >= 32 && o[l] == 0.0) zero_loads++; }
    printf("soffset %u: in-range loads correct %d/32, out-of-range loads returning 0: %d/32, cells changed inside %d (expect 32), cells changed OUTSIDE num_records %d (expect 0)\n",
           soff, good_loads, zero_loads, changed_in, changed_out);
  }
  return 0;
}
