// fill_bw.hip -- what does a pure zero-fill of the C2 canvas (12 x 1440 x 1440 cells x 128 B = 3.19 GB) sustain on one MI355X, by
// store pattern?  The reader's fused PFN + fill launch is bound by exactly this: 5.5 TB/s measured in the product.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/microbench/fill_bw tools/microbench/fill_bw.hip      Run: tools/microbench/fill_bw [frames]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x)                                                                  \
  do {                                                                         \
    hipError_t e_ = (x);                                                       \
    if (e_ != hipSuccess) {                                                    \
      fprintf(stderr, "%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); \
      exit(1);                                                                 \
    }                                                                          \
  } while (0)

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <bool NT>
__device__ __forceinline__ void st16(uint4* p) {
  if (NT) __builtin_nontemporal_store(u32x4{0u, 0u, 0u, 0u}, reinterpret_cast<u32x4*>(p));
  else *p = make_uint4(0, 0, 0, 0);
}

// (a) one block per contiguous piece of `piece` 16-byte chunks
template <bool NT>
__global__ __launch_bounds__(256) void k_piece(uint4* out, int64_t n16, int piece) {
  const int64_t b0 = (int64_t)blockIdx.x * piece;
  for (int i = threadIdx.x; i < piece; i += 256)
    if (b0 + i < n16) st16<NT>(out + b0 + i);
}

// (b) persistent blocks, pieces by ticket
template <bool NT>
__global__ __launch_bounds__(256) void k_ticket(uint4* out, int64_t n16, int piece, int* counter) {
  __shared__ int s_k;
  const int64_t npieces = (n16 + piece - 1) / piece;
  for (;;) {
    if (threadIdx.x == 0) s_k = atomicAdd(counter, 1);
    __syncthreads();
    const int k = s_k;
    __syncthreads();
    if (k >= npieces) break;
    const int64_t b0 = (int64_t)k * piece;
    for (int i = threadIdx.x; i < piece; i += 256)
      if (b0 + i < n16) st16<NT>(out + b0 + i);
  }
}

// (c) persistent blocks, static round-robin deal (no atomics, no barriers)
template <bool NT>
__global__ __launch_bounds__(256) void k_static(uint4* out, int64_t n16, int piece) {
  const int64_t npieces = (n16 + piece - 1) / piece;
  for (int64_t k = blockIdx.x; k < npieces; k += gridDim.x) {
    const int64_t b0 = k * piece;
    for (int i = threadIdx.x; i < piece; i += 256)
      if (b0 + i < n16) st16<NT>(out + b0 + i);
  }
}

// (d) the product's shape: 32 x 32-cell tiles of an NHWC canvas (32 rows of 4 KiB at a pitch of gx * 128 B), a mask word per column
// loaded per tile (dependent load in front of the stores), tiles by ticket
template <bool NT>
__global__ __launch_bounds__(256) void k_tiles(uint4* out, const uint32_t* mask, int gx, int gy, int B, int* counter) {
  __shared__ uint32_t s_word[33];
  const int tiles_x = gx / 32, tiles_y = gy / 32, t = threadIdx.x;
  const int ntiles = tiles_x * tiles_y * B;
  for (;;) {
    if (t == 0) s_word[32] = (uint32_t)atomicAdd(counter, 1);
    __syncthreads();
    int tile = (int)s_word[32];
    if (tile >= ntiles) break;
    const int tx = tile % tiles_x;
    tile /= tiles_x;
    const int ty = tile % tiles_y, b = tile / tiles_y;
    if (t < 32) s_word[t] = mask[((int64_t)(b * gx + tx * 32 + t) * gy + ty * 32) >> 5];
    __syncthreads();
    for (int idx = t; idx < 32 * 32 * 8; idx += 256) {
      const int q = idx & 7, xl = (idx >> 3) & 31, yl = idx >> 8;
      if (!((s_word[xl] >> yl) & 1u)) st16<NT>(out + (((int64_t)b * gy + ty * 32 + yl) * gx + tx * 32 + xl) * 8 + q);
    }
    __syncthreads();
  }
}

// (e) the same tiles, masks fetched ONE TILE AHEAD and stores issued as unconditional buffer stores (a masked lane's offset is out of
// range, the hardware drops it): no branch, so hipcc can count the stores between a load and its use (vmcnt is in-order on gfx9)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(void* base, uint32_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(base, 0, bytes, 0x00020000);  // raw buffer, DST_SEL defaults irrelevant for raw stores
}
template <int AUX>
__global__ __launch_bounds__(256) void k_tiles_pf(uint4* out, const uint32_t* mask, int gx, int gy, int B) {
  __shared__ uint32_t s_word[2][32];
  const int tiles_x = gx / 32, tiles_y = gy / 32, t = threadIdx.x;
  const int ntiles = tiles_x * tiles_y * B;
  auto mask_of = [&](int tile) -> uint32_t {
    const int tx = tile % tiles_x;
    tile /= tiles_x;
    const int ty = tile % tiles_y, b = tile / tiles_y;
    return mask[((int64_t)(b * gx + tx * 32 + (t & 31)) * gy + ty * 32) >> 5];
  };
  int tile = blockIdx.x;
  uint32_t m = tile < ntiles ? mask_of(tile) : 0u;
  int buf = 0;
  for (; tile < ntiles; tile += gridDim.x, buf ^= 1) {
    if (t < 32) s_word[buf][t] = m;
    __syncthreads();
    const int nxt = tile + gridDim.x;
    if (nxt < ntiles) m = mask_of(nxt);  // in flight under this tile's stores
    int tl = tile;
    const int tx = tl % tiles_x;
    tl /= tiles_x;
    const int ty = tl % tiles_y, b = tl / tiles_y;
    char* base = reinterpret_cast<char*>(out + (((int64_t)b * gy + ty * 32) * gx + tx * 32) * 8);
    const uint32_t pitch = (uint32_t)gx * 128u;
    __amdgpu_buffer_rsrc_t rs = make_rsrc(base, 31u * pitch + 4096u);
#pragma unroll
    for (int it = 0; it < 32; it++) {
      const int idx = it * 256 + t;
      const int q = idx & 7, xl = (idx >> 3) & 31, yl = idx >> 8;
      const bool occ = (s_word[buf][xl] >> yl) & 1u;
      const uint32_t off = occ ? 0xFFFFFFF0u : (uint32_t)yl * pitch + (uint32_t)(xl * 8 + q) * 16u;
      __builtin_amdgcn_raw_buffer_store_b128(u32x4{0u, 0u, 0u, 0u}, rs, off, 0, AUX);
    }
  }
}

// (f) one wave writes whole 128-byte lines with fewer, wider instructions?  (dwordx4 is the widest store: this variant makes a lane
// write 4 consecutive chunks = 64 B, i.e. 4 store instructions per lane to one line-half, to see whether line-contiguity per lane matters)
template <bool NT>
__global__ __launch_bounds__(256) void k_lane64(uint4* out, int64_t n16) {
  const int64_t per = 4;
  for (int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * per; i < n16; i += (int64_t)gridDim.x * 256 * per)
#pragma unroll
    for (int k = 0; k < 4; k++) st16<NT>(out + i + k);
}

int main(int argc, char** argv) {
  const int frames = argc > 1 ? atoi(argv[1]) : 12;
  const int gx = 1440, gy = 1440;
  const int64_t bytes = (int64_t)frames * gx * gy * 128, n16 = bytes / 16;
  uint4* buf;
  CK(hipMalloc(&buf, bytes));
  int* counter;
  CK(hipMalloc(&counter, 4));
  uint32_t* mask;
  const int64_t mwords = (int64_t)frames * gx * gy / 32;
  CK(hipMalloc(&mask, mwords * 4));
  {  // ~5 % occupied cells
    uint32_t* h = (uint32_t*)malloc(mwords * 4);
    uint64_t s = 12345;
    for (int64_t i = 0; i < mwords; i++) {
      uint32_t w = 0;
      for (int k = 0; k < 32; k++) {
        s = s * 6364136223846793005ull + 1442695040888963407ull;
        if ((s >> 33) % 100 < 5) w |= 1u << k;
      }
      h[i] = w;
    }
    CK(hipMemcpy(mask, h, mwords * 4, hipMemcpyHostToDevice));
    free(h);
  }
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  hipStream_t st;
  CK(hipStreamCreate(&st));
  auto run = [&](const char* name, auto&& launch) {
    float best = 1e30f, sum = 0;
    const int reps = 6;
    for (int r = 0; r < reps; r++) {
      CK(hipMemsetAsync(counter, 0, 4, st));
      CK(hipEventRecord(e0, st));
      launch();
      CK(hipEventRecord(e1, st));
      CK(hipEventSynchronize(e1));
      CK(hipGetLastError());
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      if (r > 0) {
        sum += ms;
        if (ms < best) best = ms;
      }
    }
    const float avg = sum / (reps - 1);
    printf("%-44s avg %8.1f us  best %8.1f us   %.2f TB/s (best %.2f)\n", name, avg * 1e3, best * 1e3, bytes / (avg * 1e-3) / 1e12, bytes / (best * 1e-3) / 1e12);
    fflush(stdout);
  };
  printf("# zero-fill of %d frames x %d x %d x 128 B = %.1f MB\n", frames, gx, gy, bytes / 1e6);
  run("hipMemsetAsync", [&] { CK(hipMemsetAsync(buf, 0, bytes, st)); });
  for (int piece : {1024, 8192, 65536}) {
    char nm[96];
    const int nb = (int)((n16 + piece - 1) / piece);
    snprintf(nm, sizeof nm, "block per %d KiB piece, plain", piece / 64);
    run(nm, [&] { k_piece<false><<<nb, 256, 0, st>>>(buf, n16, piece); });
    snprintf(nm, sizeof nm, "block per %d KiB piece, nontemporal", piece / 64);
    run(nm, [&] { k_piece<true><<<nb, 256, 0, st>>>(buf, n16, piece); });
  }
  for (int blocks : {256, 512, 1024, 2048}) {
    char nm[96];
    snprintf(nm, sizeof nm, "%d persistent blocks, 128 KiB by ticket, nt", blocks);
    run(nm, [&] { k_ticket<true><<<blocks, 256, 0, st>>>(buf, n16, 8192, counter); });
    snprintf(nm, sizeof nm, "%d persistent blocks, 128 KiB static, nt", blocks);
    run(nm, [&] { k_static<true><<<blocks, 256, 0, st>>>(buf, n16, 8192); });
    snprintf(nm, sizeof nm, "%d persistent blocks, 128 KiB static, plain", blocks);
    run(nm, [&] { k_static<false><<<blocks, 256, 0, st>>>(buf, n16, 8192); });
    snprintf(nm, sizeof nm, "%d persistent blocks, 16 KiB static, nt", blocks);
    run(nm, [&] { k_static<true><<<blocks, 256, 0, st>>>(buf, n16, 1024); });
  }
  for (int blocks : {256, 512, 1024}) {
    char nm[96];
    snprintf(nm, sizeof nm, "%d blocks, masked 32x32 tiles by ticket, nt", blocks);
    run(nm, [&] { k_tiles<true><<<blocks, 256, 0, st>>>(buf, mask, gx, gy, frames, counter); });
    snprintf(nm, sizeof nm, "%d blocks, masked tiles, plain", blocks);
    run(nm, [&] { k_tiles<false><<<blocks, 256, 0, st>>>(buf, mask, gx, gy, frames, counter); });
    snprintf(nm, sizeof nm, "%d blocks, masked tiles prefetched, buffer nt", blocks);
    run(nm, [&] { k_tiles_pf<2><<<blocks, 256, 0, st>>>(buf, mask, gx, gy, frames); });
    snprintf(nm, sizeof nm, "%d blocks, masked tiles prefetched, buffer plain", blocks);
    run(nm, [&] { k_tiles_pf<0><<<blocks, 256, 0, st>>>(buf, mask, gx, gy, frames); });
  }
  run("grid-stride 64 B per lane, 4096 blocks, nt", [&] { k_lane64<true><<<4096, 256, 0, st>>>(buf, n16); });
  run("grid-stride 64 B per lane, 4096 blocks, plain", [&] { k_lane64<false><<<4096, 256, 0, st>>>(buf, n16); });
  return 0;
}
