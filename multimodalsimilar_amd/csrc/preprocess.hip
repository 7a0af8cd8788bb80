// Image input stage on the GPU: the reference's timm / torchvision / Pillow transform ahead of the image
// tower (multimodal_infer.py:86-91: create_transform(input_size, 'bicubic', mean, std, crop_pct) = Resize -> CenterCrop ->
// ToTensor -> Normalize), one launch pair per image, bit-exact with Pillow's 8-bit two-pass resampling:
//   pass 1  horizontal: tmp[y][x'][c] = clip8((2^21 + sum_t kx[x'][t] * img[y][x0(x') + t][c]) >> 22)     (Resample.c *_8bpc)
//   pass 2  vertical + crop + ToTensor + Normalize: out[c][y'][x'] = ((clip8(...) / 255) - mean[c]) / std[c], fp32, IEEE division
// Only what the crop window needs is computed: pass 1 covers the window's columns and the input rows the window's vertical
// taps reach.  The fixed-point kernels (22 fractional bits, Pillow's rounding) are built on the host (preprocess.py) and
// cached per (input size, output size); HBM-bound byte work -- uint8 in, fp32 out, no MFMA.
#include "common.h"

struct PrepGeom {
  int H, W, pitch;             // input image [H][W][3] uint8, row pitch in bytes
  int ksx, ksy;                // taps per output column / row in the coefficient tables
  int left, top, S;            // crop window in the resized image; output is [3][S][S]
  int row0, nrows;             // input rows pass 1 covers: [row0, row0 + nrows)
};

__device__ __forceinline__ int clip8(int v) { return min(max(v >> 22, 0), 255); }

// thread = (tmp row, output column of the window); tmp is [nrows][S][4] bytes (rgb + pad: one 4-byte store)
__global__ __launch_bounds__(256) void prep_h_kernel(const unsigned char* __restrict__ img, const int* __restrict__ bx,
                                                     const int* __restrict__ kx, unsigned char* __restrict__ tmp, PrepGeom g) {
  const int x = blockIdx.x * 64 + (threadIdx.x & 63), r = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (x >= g.S || r >= g.nrows) return;
  const int xo = g.left + x;
  const int x0 = bx[2 * xo], n = bx[2 * xo + 1];
  const int* k = kx + (size_t)xo * g.ksx;
  const unsigned char* src = img + (size_t)(g.row0 + r) * g.pitch + (size_t)x0 * 3;
  int a0 = 1 << 21, a1 = 1 << 21, a2 = 1 << 21;
  for (int t = 0; t < n; ++t) {
    const int c = k[t];
    a0 += c * src[3 * t]; a1 += c * src[3 * t + 1]; a2 += c * src[3 * t + 2];
  }
  const unsigned int o = (unsigned int)clip8(a0) | ((unsigned int)clip8(a1) << 8) | ((unsigned int)clip8(a2) << 16);
  reinterpret_cast<unsigned int*>(tmp)[(size_t)r * g.S + x] = o;
}

// thread = (output row, output column); consecutive lanes = consecutive columns (coalesced tmp reads and fp32 stores)
__global__ __launch_bounds__(256) void prep_v_kernel(const unsigned char* __restrict__ tmp, const int* __restrict__ by,
                                                     const int* __restrict__ ky, float* __restrict__ out, PrepGeom g,
                                                     float m0, float m1, float m2, float s0, float s1, float s2) {
  const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (x >= g.S || y >= g.S) return;
  const int yo = g.top + y;
  const int y0 = by[2 * yo], n = by[2 * yo + 1];
  const int* k = ky + (size_t)yo * g.ksy;
  const unsigned int* src = reinterpret_cast<const unsigned int*>(tmp) + (size_t)(y0 - g.row0) * g.S + x;
  int a0 = 1 << 21, a1 = 1 << 21, a2 = 1 << 21;
  for (int t = 0; t < n; ++t) {
    const int c = k[t];
    const unsigned int v = src[(size_t)t * g.S];
    a0 += c * (int)(v & 255u); a1 += c * (int)((v >> 8) & 255u); a2 += c * (int)((v >> 16) & 255u);
  }
  const size_t plane = (size_t)g.S * g.S, o = (size_t)y * g.S + x;
  // ToTensor: uint8 -> float / 255; Normalize: (t - mean) / std -- correctly rounded divisions, as torch computes them
  out[o] = __fdiv_rn(__fdiv_rn((float)clip8(a0), 255.0f) - m0, s0);
  out[plane + o] = __fdiv_rn(__fdiv_rn((float)clip8(a1), 255.0f) - m1, s1);
  out[2 * plane + o] = __fdiv_rn(__fdiv_rn((float)clip8(a2), 255.0f) - m2, s2);
}

extern "C" int mmsim_preprocess_image(const void* img, int H, int W, int pitch, const int* bx, const int* kx, int ksx, int out_w,
                                      const int* by, const int* ky, int ksy, int out_h, int left, int top, int S, int row0, int nrows,
                                      void* tmp, unsigned long long tmp_bytes, float* out, float mean0, float mean1, float mean2,
                                      float std0, float std1, float std2, void* stream) {
  MMSIM_REQUIRE(img && bx && kx && by && ky && tmp && out, "preprocess_image: null operand");
  MMSIM_REQUIRE(H > 0 && W > 0 && pitch >= 3 * W && ksx > 0 && ksy > 0 && S > 0, "preprocess_image: bad geometry");
  MMSIM_REQUIRE(left >= 0 && top >= 0 && left + S <= out_w && top + S <= out_h, "preprocess_image: crop window outside the resized image");
  MMSIM_REQUIRE(row0 >= 0 && nrows > 0 && row0 + nrows <= H, "preprocess_image: pass-1 row range outside the image");
  MMSIM_REQUIRE(tmp_bytes >= (unsigned long long)nrows * S * 4, "preprocess_image: tmp too small (nrows * S * 4 bytes)");
  MMSIM_REQUIRE(std0 != 0.f && std1 != 0.f && std2 != 0.f, "preprocess_image: std must be non-zero");
  PrepGeom g; g.H = H; g.W = W; g.pitch = pitch; g.ksx = ksx; g.ksy = ksy; g.left = left; g.top = top; g.S = S; g.row0 = row0; g.nrows = nrows;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(prep_h_kernel, dim3((S + 63) / 64, (nrows + 3) / 4), dim3(256), 0, s, (const unsigned char*)img, bx, kx, (unsigned char*)tmp, g);
  hipLaunchKernelGGL(prep_v_kernel, dim3((S + 63) / 64, (S + 3) / 4), dim3(256), 0, s, (const unsigned char*)tmp, by, ky, out, g,
                     mean0, mean1, mean2, std0, std1, std2);
  return mmsim_check_launch("preprocess_image");
}
