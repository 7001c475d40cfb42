// Device-side tail of the input pipeline: uint8 HWC frames -> cropped, flipped, mean/std-normalised NCDHW clips.
//
// The reference builds every training sample on the host (lib/data/datasets/video_contrast_dataset.py:174-203 get_item: two
// augmented views of T frames, concatenated on the channel axis) and ships 154 MB of fp32 per 32-clip batch over PCIe.  The
// last three stages of its transform chain (lib/data/transform/build.py:45-62) are pure data movement + one affine map:
//
//   VideoRandomHorizontalFlip   consistency_transforms.py  (cv2.flip(img, 1): reverse W, same decision for all T frames)
//   VideoNormalize              consistency_transforms.py:45-65  img.astype(f32); img -= mean*255; img *= 1/(std*255)
//   VideoToTensor               consistency_transforms.py:11-43  stack T frames (H,W,C,T) -> (C,T,H,W) float
//
// plus an integer crop window (albumentations' random_crop coordinates, as VideoRandomCrop / VideoCenterCrop use them).  Here
// the host hands over the uint8 frames (4x fewer bytes over PCIe) and ONE pass over HBM does crop + flip + normalise + the
// HWC -> CDHW layout change + the concatenation of the two views into the (b, 6, T, H, W) batch the trainer consumes, in
// fp32 or (fp16-storage path) fp16.  Arithmetic is exactly the reference's: (float(px) - m_c) * d_c with two fp32 roundings,
// m_c = f32(mean_c) * 255 and d_c = 1 / (f32(std_c) * 255) computed on the host in fp32 the way numpy does -- bit-exact.
// HBM-bound: 3 B read + 12 B (6 B) written per pixel.
#include "gca_common.h"

namespace {

struct ClipParams {
  int views, T, Hs, Ws, H, W;
  long long total4;             // threads: b * views * T * H * ceil(W / 4)
  int w4;                       // ceil(W / 4)
  float m[3], d[3];
};

// One thread: 4 consecutive output columns of one (clip, view, frame, row), all three channels.  The 12 source bytes are
// contiguous (HWC); a flipped row reads the mirrored 12 bytes and reverses them in registers.
template <typename T, bool VEC>
__global__ __launch_bounds__(256) void clip_prepare_kernel(const unsigned char* __restrict__ frames,
                                                           const int* __restrict__ params, T* __restrict__ out,
                                                           const ClipParams p) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= p.total4) return;
  const int wq = (int)(i % p.w4);
  long long r = i / p.w4;
  const int h = (int)(r % p.H); r /= p.H;
  const int t = (int)(r % p.T); r /= p.T;
  const int v = (int)(r % p.views);
  const long long n = r / p.views;
  const int* pr = params + (n * p.views + v) * 4;
  // (the window is clamped into the frame: a bad table can mis-crop, never read outside the buffer)
  const int h0 = min(max(pr[0], 0), p.Hs - p.H), w0 = min(max(pr[1], 0), p.Ws - p.W), flip = pr[2];
  const int w = wq * 4;
  const int nw = min(4, p.W - w);
  // source row of this (clip, view, frame): (Hs, Ws, 3) bytes
  const unsigned char* row = frames + ((((n * p.views + v) * p.T + t) * (long long)p.Hs + (h0 + h)) * p.Ws) * 3;
  float px[4][3];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    // output column w + j reads source column w0 + (w + j), or its mirror inside the crop window
    const int wo = w + (j < nw ? j : 0);
    const int ws = w0 + (flip ? p.W - 1 - wo : wo);
#pragma unroll
    for (int c = 0; c < 3; ++c) px[j][c] = (float)row[ws * 3 + c];
  }
  const long long plane = (long long)p.T * p.H * p.W;
  T* o = out + ((n * p.views + v) * 3) * plane + ((long long)t * p.H + h) * p.W + w;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    float y[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float s = px[j][c] - p.m[c];           // img -= mean   (one rounding)
      y[j] = s * p.d[c];                           // img *= 1/std  (one rounding; never contracted into an fma of px)
    }
    if (VEC) gca_act<T>::st4(o + c * plane, make_float4(y[0], y[1], y[2], y[3]));
    else
      for (int j = 0; j < nw; ++j) gca_act<T>::st(o + c * plane + j, y[j]);
  }
}

}  // namespace

extern "C" {

int gca_clip_prepare(const uint8_t* frames, int64_t b, int64_t views, int64_t T, int64_t Hs, int64_t Ws,
                     const int32_t* params, const float* mean255, const float* inv_std255, int64_t H, int64_t W,
                     void* out, int out_f16, void* stream) {
  if (!frames || !params || !mean255 || !inv_std255 || !out || b <= 0 || views <= 0 || views > 8 || T <= 0 || Hs <= 0 ||
      Ws <= 0 || H <= 0 || W <= 0 || H > Hs || W > Ws || Hs > 32767 || Ws > 32767 || T > 32767)
    return GCA_EINVAL;
  ClipParams p;
  p.views = (int)views; p.T = (int)T; p.Hs = (int)Hs; p.Ws = (int)Ws; p.H = (int)H; p.W = (int)W;
  p.w4 = (int)((W + 3) / 4);
  p.total4 = b * views * T * H * p.w4;
  for (int c = 0; c < 3; ++c) { p.m[c] = mean255[c]; p.d[c] = inv_std255[c]; }      // HOST pointers: six floats by value
  const long long blocks = gca_ceil_div(p.total4, 256);
  if (blocks > 0x7fffffffLL) return GCA_EINVAL;
  const bool vec = W % 4 == 0 && ((uintptr_t)out % 16) == 0;
  const dim3 grid((unsigned)blocks);
  hipStream_t st = (hipStream_t)stream;
  const unsigned char* f = frames;
  if (out_f16) {
    if (vec) hipLaunchKernelGGL((clip_prepare_kernel<gca_half, true>), grid, dim3(256), 0, st, f, params, (gca_half*)out, p);
    else hipLaunchKernelGGL((clip_prepare_kernel<gca_half, false>), grid, dim3(256), 0, st, f, params, (gca_half*)out, p);
  } else {
    if (vec) hipLaunchKernelGGL((clip_prepare_kernel<float, true>), grid, dim3(256), 0, st, f, params, (float*)out, p);
    else hipLaunchKernelGGL((clip_prepare_kernel<float, false>), grid, dim3(256), 0, st, f, params, (float*)out, p);
  }
  return gca_launch_status();
}

}  // extern "C"
