// titanet_amd — mel front end on the GPU (reference src/transforms.py:158-203, torchaudio 0.13 semantics:
// Spectrogram(n_fft, win_length, hop, power=None, center, reflect) -> |.|^2 -> MelScale(HTK, norm=None)
// -> AmplitudeToDB(power, amin 1e-10, ref 1) -> F.normalize over the mel axis -> SpecAugment masks).
// One workgroup per frame: windowed frame -> radix-2 FFT in LDS -> power -> sparse mel triangles -> dB
// -> L2 normalisation -> masked store in the [B, n_mels, T] layout TitaNet.forward consumes.
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "../../include/titanet_amd.h"
#include "tn_common.h"

struct tn_mel {
  int sample_rate, n_fft, win_length, hop, n_mels, log2n, n_freqs;
  float* window = nullptr;   // [n_fft] hann (periodic) centred in n_fft
  float* fb = nullptr;       // [n_mels][n_freqs]
  int* range = nullptr;      // [n_mels][2] first / one-past-last non-zero bin
  float* twiddle = nullptr;  // [n_fft/2][2] cos, -sin of 2 pi k / n_fft
  // n_fft = 512 (the reference's front end, parameters.yml): the one-wave-per-frame kernel's tables (mel512_batch_kernel)
  bool fast512 = false;
  float* tw512 = nullptr;    // [8][64][2]  W_512^(k1 n2)
  float* tw64 = nullptr;     // [8][8][2]   W_64^(j1 m2)
  float* sp_w = nullptr;     // the non-zero filterbank weights, mel after mel (sp_off[m] .. sp_off[m + 1])
  int* sp_off = nullptr;     // [n_mels + 1]
  int nnz = 0;
};

__global__ void mel_frame_kernel(const float* __restrict__ waves, int64_t n_samples, int T, int n_fft, int log2n, int hop,
                                 int n_mels, int n_freqs, const float* __restrict__ window, const float* __restrict__ fb,
                                 const int* __restrict__ range, const float* __restrict__ twiddle,
                                 const int32_t* __restrict__ masks, float* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* re = reinterpret_cast<float*>(smem);
  float* im = re + n_fft;
  float* pw = im + n_fft;            // [n_freqs]
  float* mel = pw + n_freqs + 3;     // [n_mels]
  __shared__ float s_norm;
  const int t = blockIdx.x, b = blockIdx.y, tid = threadIdx.x, half_n = n_fft >> 1;
  const float* wave = waves + (size_t)b * n_samples;
  // ---- windowed frame (center=True, reflect padding), bit-reversed order
  for (int n = tid; n < n_fft; n += blockDim.x) {
    int64_t s = (int64_t)t * hop - half_n + n;
    if (s < 0) s = -s;
    if (s >= n_samples) s = 2 * (n_samples - 1) - s;
    if (s < 0) s = 0;   // degenerate: utterance shorter than the padding
    const unsigned r = __brev((unsigned)n) >> (32 - log2n);
    re[r] = wave[s] * window[n];
    im[r] = 0.f;
  }
  __syncthreads();
  // ---- radix-2 decimation-in-time FFT
  for (int st = 0; st < log2n; ++st) {
    const int half = 1 << st, len = half << 1;
    for (int j = tid; j < half_n; j += blockDim.x) {
      const int grp = j >> st, pos = j & (half - 1);
      const int i0 = grp * len + pos, i1 = i0 + half;
      const int tw = pos << (log2n - 1 - st);
      const float c = twiddle[2 * tw], s = twiddle[2 * tw + 1];
      const float xr = re[i1] * c - im[i1] * s, xi = re[i1] * s + im[i1] * c;
      const float ar = re[i0], ai = im[i0];
      re[i0] = ar + xr; im[i0] = ai + xi;
      re[i1] = ar - xr; im[i1] = ai - xi;
    }
    __syncthreads();
  }
  for (int k = tid; k < n_freqs; k += blockDim.x) pw[k] = re[k] * re[k] + im[k] * im[k];
  __syncthreads();
  // ---- mel triangles (sparse), dB
  for (int m = tid; m < n_mels; m += blockDim.x) {
    float s = 0.f;
    const int lo = range[2 * m], hi = range[2 * m + 1];
    for (int k = lo; k < hi; ++k) s = fmaf(fb[(size_t)m * n_freqs + k], pw[k], s);
    mel[m] = 10.f * log10f(fmaxf(s, 1e-10f));
  }
  __syncthreads();
  if (tid < 64) {
    float q = 0.f;
    for (int m = tid; m < n_mels; m += 64) q += mel[m] * mel[m];
    q = wave_sum(q);
    if (tid == 0) s_norm = fmaxf(sqrtf(q), 1e-12f);
  }
  __syncthreads();
  int f0 = 0, f1 = 0, t0 = 0, t1 = 0;
  if (masks) { f0 = masks[4 * b]; f1 = masks[4 * b + 1]; t0 = masks[4 * b + 2]; t1 = masks[4 * b + 3]; }
  const bool tmask = t >= t0 && t < t1;
  const float inv = 1.f / s_norm;
  for (int m = tid; m < n_mels; m += blockDim.x) {
    float v = mel[m] * inv;
    if (tmask || (m >= f0 && m < f1)) v = 0.f;
    out[((size_t)b * n_mels + m) * T + t] = v;
  }
}

// ------------------------------------------------------------------------------------------
// Batched form (BASELINE.json configs[3]: variable-length utterances, SpecAugment on the GPU): utterance b has lengths[b]
// samples (its frames, its reflect padding), an optional time-stretch rate and boolean frequency / time masks (any number
// of mask_along_axis intervals folded into one byte vector per axis).  The reference takes .abs().pow(2) right after the
// phase vocoder (src/transforms.py:173-177), so only the vocoder's magnitude path matters:
//   |out[j]| = alpha |S[idx + 1]| + (1 - alpha) |S[idx]|,  j * rate = idx + alpha,  S zero beyond its last frame.
// One workgroup produces FT = 16 consecutive frames and stores them through an LDS tile, so the [B, n_mels, T] output
// is written in 64-byte runs along T instead of one float per (mel, frame) at stride T.
// ------------------------------------------------------------------------------------------
#define MEL_FT 16
__device__ void mel_fft_magnitude(const float* __restrict__ wave, int64_t len, int frame, int n_frames, int n_fft, int log2n, int hop,
                                  int n_freqs, const float* __restrict__ window, const float* __restrict__ twiddle, float* re, float* im,
                                  float* mag, bool squared) {
  const int tid = threadIdx.x, half_n = n_fft >> 1;
  if (frame >= n_frames) {       // beyond the utterance: the vocoder's zero padding
    for (int k = tid; k < n_freqs; k += blockDim.x) mag[k] = 0.f;
    __syncthreads();
    return;
  }
  for (int n = tid; n < n_fft; n += blockDim.x) {
    int64_t s = (int64_t)frame * hop - half_n + n;
    if (s < 0) s = -s;
    if (s >= len) s = 2 * (len - 1) - s;
    if (s < 0) s = 0;
    const unsigned r = __brev((unsigned)n) >> (32 - log2n);
    re[r] = wave[s] * window[n];
    im[r] = 0.f;
  }
  __syncthreads();
  for (int st = 0; st < log2n; ++st) {
    const int half = 1 << st, len2 = half << 1;
    for (int j = tid; j < half_n; j += blockDim.x) {
      const int grp = j >> st, pos = j & (half - 1);
      const int i0 = grp * len2 + pos, i1 = i0 + half;
      const int tw = pos << (log2n - 1 - st);
      const float c = twiddle[2 * tw], sn = twiddle[2 * tw + 1];
      const float xr = re[i1] * c - im[i1] * sn, xi = re[i1] * sn + im[i1] * c;
      const float ar = re[i0], ai = im[i0];
      re[i0] = ar + xr; im[i0] = ai + xi;
      re[i1] = ar - xr; im[i1] = ai - xi;
    }
    __syncthreads();
  }
  for (int k = tid; k < n_freqs; k += blockDim.x) {
    const float p = re[k] * re[k] + im[k] * im[k];
    mag[k] = squared ? p : sqrtf(p);
  }
  __syncthreads();
}

__global__ void mel_batch_kernel(const float* __restrict__ waves, int64_t n_samples_max, const int64_t* __restrict__ lengths,
                                 const double* __restrict__ rates, const uint8_t* __restrict__ fmask, const uint8_t* __restrict__ tmask,
                                 int T_out, int n_fft, int log2n, int hop, int n_mels, int n_freqs, const float* __restrict__ window,
                                 const float* __restrict__ fb, const int* __restrict__ range, const float* __restrict__ twiddle,
                                 float* __restrict__ out, unsigned short* __restrict__ packed) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* re = reinterpret_cast<float*>(smem);
  float* im = re + n_fft;
  float* p0 = im + n_fft;             // [n_freqs + 3]
  float* p1 = p0 + n_freqs + 3;       // [n_freqs + 3]
  float* mel = p1 + n_freqs + 3;      // [n_mels]
  float* tile = mel + n_mels;         // [n_mels][MEL_FT + 1]
  __shared__ float s_norm;
  const int b = blockIdx.y, tid = threadIdx.x, j0 = blockIdx.x * MEL_FT;
  const float* wave = waves + (size_t)b * n_samples_max;
  const int64_t len = lengths ? lengths[b] : n_samples_max;
  const int n_frames = (int)(1 + len / hop);
  const double rate = rates ? rates[b] : 1.0;
  int n_out = n_frames;
  if (rate != 1.0) n_out = (int)ceil((double)n_frames / rate);     // len(arange(0, n_frames, rate))
  for (int jj = 0; jj < MEL_FT; ++jj) {
    const int j = j0 + jj;
    if (j >= n_out || j >= T_out) {
      for (int m = tid; m < n_mels; m += blockDim.x) tile[m * (MEL_FT + 1) + jj] = 0.f;
      continue;       // uniform over the workgroup
    }
    if (rate == 1.0) {
      mel_fft_magnitude(wave, len, j, n_frames, n_fft, log2n, hop, n_freqs, window, twiddle, re, im, p0, true);
    } else {
      const double pos = (double)j * rate;
      const int idx = (int)floor(pos);
      const float alpha = (float)(pos - (double)idx);
      mel_fft_magnitude(wave, len, idx, n_frames, n_fft, log2n, hop, n_freqs, window, twiddle, re, im, p0, false);
      mel_fft_magnitude(wave, len, idx + 1, n_frames, n_fft, log2n, hop, n_freqs, window, twiddle, re, im, p1, false);
      for (int k = tid; k < n_freqs; k += blockDim.x) {
        const float v = alpha * p1[k] + (1.f - alpha) * p0[k];
        p0[k] = v * v;
      }
      __syncthreads();
    }
    for (int m = tid; m < n_mels; m += blockDim.x) {
      float s = 0.f;
      const int lo = range[2 * m], hi = range[2 * m + 1];
      for (int k = lo; k < hi; ++k) s = fmaf(fb[(size_t)m * n_freqs + k], p0[k], s);
      mel[m] = 10.f * log10f(fmaxf(s, 1e-10f));
    }
    __syncthreads();
    if (tid < 64) {
      float q = 0.f;
      for (int m = tid; m < n_mels; m += 64) q += mel[m] * mel[m];
      q = wave_sum(q);
      if (tid == 0) s_norm = fmaxf(sqrtf(q), 1e-12f);
    }
    __syncthreads();
    const bool tm = tmask && tmask[(size_t)b * T_out + j];
    const float inv = 1.f / s_norm;
    for (int m = tid; m < n_mels; m += blockDim.x) {
      float v = mel[m] * inv;
      if (tm || (fmask && fmask[(size_t)b * n_mels + m])) v = 0.f;
      tile[m * (MEL_FT + 1) + jj] = v;
    }
    __syncthreads();
  }
  __syncthreads();
  if (out) {
    for (int i = tid; i < n_mels * MEL_FT; i += blockDim.x) {
      const int m = i / MEL_FT, jj = i % MEL_FT;
      if (j0 + jj < T_out) out[((size_t)b * n_mels + m) * T_out + j0 + jj] = tile[m * (MEL_FT + 1) + jj];
    }
  }
  if (packed) {
    // the prolog conv's operand layout: rows x n_mels bf16, row = b * T_out + frame (what prolog_pack_kernel would make of
    // `out`): the MEL_FT frames of this workgroup are one contiguous run
    unsigned short* pb = packed + ((size_t)b * T_out + j0) * n_mels;
    const int nfr = min(MEL_FT, T_out - j0);
    for (int i = tid; i < nfr * n_mels; i += blockDim.x) {
      const int jj = i / n_mels, m = i - jj * n_mels;
      pb[i] = __builtin_bit_cast(unsigned short, (__bf16)tile[m * (MEL_FT + 1) + jj]);
    }
  }
}


// ------------------------------------------------------------------------------------------
// n_fft = 512: ONE WAVE PER FRAME (round 6).  mel_batch_kernel spends a 256-thread workgroup and 9 + 5 workgroup barriers
// on every 512-point frame (a radix-2 pass per barrier, 4 butterflies per thread), 16 frames one after the other: latency,
// not arithmetic (36.8 k frames of configs[3] took 425 us).  Here a wave owns a frame: 512 = 8 x 8 x 8, every lane holds
// 8 points, three radix-8 passes in registers with two exchanges through a wave-private LDS buffer (no workgroup barrier
// until the tile store), twiddles / window / filterbank in registers and LDS.  Under a time stretch the 4 consecutive
// output frames of a wave share their STFT frames (idx, idx + 1 | idx + 1, idx + 2 ...): 5 transforms instead of 8.
//   n = 64 n1 + n2, k = k1 + 8 (j1 + 8 j2):   X[k] = sum_m2 W8^(m2 j2) W64^(m2 j1) sum_m1 W8^(m1 j1) W512^(n2 k1) sum_n1 W8^(n1 k1) x[n],  n2 = 8 m1 + m2
// ------------------------------------------------------------------------------------------
#define MEL5_CS 66        // float2 pitch of one k1 row of the exchange buffer
#define MEL5_MAXW 2048    // filterbank non-zeros the LDS copy holds
#define MEL5_MP 128       // mel bins a wave's scratch row holds
__device__ __forceinline__ float2 m5_add(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 m5_sub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ float2 m5_mul(float2 a, float2 b) { return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
__device__ __forceinline__ float2 m5_mi(float2 a) { return make_float2(a.y, -a.x); }     // a * (-i)
__device__ __forceinline__ void m5_dft4(float2 c0, float2 c1, float2 c2, float2 c3, float2& x0, float2& x1, float2& x2, float2& x3) {
  const float2 e0 = m5_add(c0, c2), e1 = m5_sub(c0, c2), o0 = m5_add(c1, c3), o1 = m5_mi(m5_sub(c1, c3));
  x0 = m5_add(e0, o0); x1 = m5_add(e1, o1); x2 = m5_sub(e0, o0); x3 = m5_sub(e1, o1);
}
// forward 8-point DFT in place (decimation in frequency), natural order in and out
__device__ __forceinline__ void m5_dft8(float2 (&v)[8]) {
  const float h = 0.70710678118654752f;
  const float2 a0 = m5_add(v[0], v[4]), a1 = m5_add(v[1], v[5]), a2 = m5_add(v[2], v[6]), a3 = m5_add(v[3], v[7]);
  const float2 d0 = m5_sub(v[0], v[4]), d1 = m5_sub(v[1], v[5]), d2 = m5_sub(v[2], v[6]), d3 = m5_sub(v[3], v[7]);
  const float2 b1 = make_float2((d1.x + d1.y) * h, (d1.y - d1.x) * h);        // d1 (1 - i) / sqrt 2
  const float2 b2 = m5_mi(d2);
  const float2 b3 = make_float2((d3.y - d3.x) * h, -(d3.x + d3.y) * h);       // d3 (-1 - i) / sqrt 2
  m5_dft4(a0, a1, a2, a3, v[0], v[2], v[4], v[6]);
  m5_dft4(d0, b1, b2, b3, v[1], v[3], v[5], v[7]);
}
__device__ __forceinline__ void m5_wave_sync() {      // LDS exchange inside ONE wave: its LDS operations complete in order
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}
struct Mel5Lane { float win[8]; float2 t512[8]; float2 t64[8]; };
// |STFT frame|^2 (squared) or |.| of one frame -> mag[0 .. 256]; frames beyond the utterance: the vocoder's zero padding
__device__ __forceinline__ void m5_fft(const float* __restrict__ wave, int64_t len, int frame, int n_frames, int hop, const Mel5Lane& L,
                                       float2* cb, float* mag, bool squared, int lane) {
  if (frame >= n_frames) {
    for (int k = lane; k < 257; k += 64) mag[k] = 0.f;
    m5_wave_sync();
    return;
  }
  float2 v[8];
  const int64_t s0 = (int64_t)frame * hop - 256 + lane;
  if (s0 >= 0 && s0 + 448 < len) {
#pragma unroll
    for (int n1 = 0; n1 < 8; ++n1) v[n1] = make_float2(wave[s0 + 64 * n1] * L.win[n1], 0.f);
  } else {
#pragma unroll
    for (int n1 = 0; n1 < 8; ++n1) {
      int64_t s = s0 + 64 * n1;
      if (s < 0) s = -s;
      if (s >= len) s = 2 * (len - 1) - s;
      if (s < 0) s = 0;       // degenerate: utterance shorter than the padding
      v[n1] = make_float2(wave[s] * L.win[n1], 0.f);
    }
  }
  m5_dft8(v);
#pragma unroll
  for (int k1 = 0; k1 < 8; ++k1) cb[k1 * MEL5_CS + lane] = m5_mul(v[k1], L.t512[k1]);
  m5_wave_sync();
  float2* row = cb + (lane >> 3) * MEL5_CS;
#pragma unroll
  for (int m1 = 0; m1 < 8; ++m1) v[m1] = row[8 * m1 + (lane & 7)];
  m5_wave_sync();
  m5_dft8(v);
#pragma unroll
  for (int j1 = 0; j1 < 8; ++j1) row[j1 * 8 + (lane & 7)] = m5_mul(v[j1], L.t64[j1]);
  m5_wave_sync();
#pragma unroll
  for (int m2 = 0; m2 < 8; m2 += 2) {
    const float4 t = *reinterpret_cast<const float4*>(row + (lane & 7) * 8 + m2);
    v[m2] = make_float2(t.x, t.y); v[m2 + 1] = make_float2(t.z, t.w);
  }
  m5_dft8(v);
  const int kb = (lane >> 3) + 8 * (lane & 7);
#pragma unroll
  for (int j2 = 0; j2 < 4; ++j2) {
    const float p = v[j2].x * v[j2].x + v[j2].y * v[j2].y;
    mag[kb + 64 * j2] = squared ? p : sqrtf(p);
  }
  if (lane == 0) {
    const float p = v[4].x * v[4].x + v[4].y * v[4].y;
    mag[256] = squared ? p : sqrtf(p);
  }
  m5_wave_sync();
}

__global__ __launch_bounds__(256) void mel512_batch_kernel(const float* __restrict__ waves, int64_t n_samples_max, const int64_t* __restrict__ lengths,
                                                           const double* __restrict__ rates, const uint8_t* __restrict__ fmask,
                                                           const uint8_t* __restrict__ tmask, int T_out, int hop, int n_mels,
                                                           const float* __restrict__ window, const float2* __restrict__ tw512,
                                                           const float2* __restrict__ tw64, const float* __restrict__ sp_w,
                                                           const int* __restrict__ sp_off, const int* __restrict__ range, int nnz,
                                                           const int32_t* __restrict__ imasks, float* __restrict__ out,
                                                           unsigned short* __restrict__ packed) {
  // imasks (tn_mel_forward): int32 [B][4] = one frequency and one time interval per utterance, or null
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float2* cb_all = reinterpret_cast<float2*>(smem);                         // [4 waves][8][MEL5_CS]
  float* mag_all = reinterpret_cast<float*>(cb_all + 4 * 8 * MEL5_CS);      // [4][2][264]
  float* melv_all = mag_all + 4 * 2 * 264;                                  // [4][MEL5_MP]
  float* tile = melv_all + 4 * MEL5_MP;                                     // [n_mels][MEL_FT + 1]
  float* wts = tile + n_mels * (MEL_FT + 1);                                // [nnz]
  const int b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, j0 = blockIdx.x * MEL_FT;
  const float* wave = waves + (size_t)b * n_samples_max;
  const int64_t len = lengths ? lengths[b] : n_samples_max;
  const int n_frames = (int)(1 + len / hop);
  const double rate = rates ? rates[b] : 1.0;
  int n_out = n_frames;
  if (rate != 1.0) n_out = (int)ceil((double)n_frames / rate);     // len(arange(0, n_frames, rate))
  float2* cb = cb_all + wv * 8 * MEL5_CS;
  float* mag0 = mag_all + wv * 2 * 264;
  float* melv = melv_all + wv * MEL5_MP;
  if (j0 < n_out) {
    for (int i = tid; i < nnz; i += 256) wts[i] = sp_w[i];
  }
  __syncthreads();
  if (j0 < n_out) {
    Mel5Lane L;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      L.win[i] = window[64 * i + lane];
      L.t512[i] = tw512[i * 64 + lane];
      L.t64[i] = tw64[i * 8 + (lane & 7)];
    }
    // this lane's filterbank rows: mel `lane` whole, and a quarter of mel 64 + 16 s + (lane >> 2) per sub-pass s
    int loA = 0, cntA = 0, offA = 0;
    if (lane < n_mels) { loA = range[2 * lane]; cntA = range[2 * lane + 1] - loA; offA = sp_off[lane]; }
    int if0 = 0, if1 = 0, it0 = 0, it1 = 0;
    if (imasks) { if0 = imasks[4 * b]; if1 = imasks[4 * b + 1]; it0 = imasks[4 * b + 2]; it1 = imasks[4 * b + 3]; }
    int tag0 = -1, tag1 = -1;    // STFT frames whose magnitudes mag0, mag0 + 264 hold (time stretch)
    for (int q = 0; q < 4; ++q) {
      const int jj = wv * 4 + q, j = j0 + jj;
      if (j >= n_out || j >= T_out) {
        for (int m = lane; m < n_mels; m += 64) tile[m * (MEL_FT + 1) + jj] = 0.f;
        continue;       // uniform over the wave
      }
      const float* pw = mag0;
      if (rate == 1.0) {
        m5_fft(wave, len, j, n_frames, hop, L, cb, mag0, true, lane);
      } else {
        const double pos = (double)j * rate;
        const int idx = (int)floor(pos);
        const float alpha = (float)(pos - (double)idx);
        if (tag0 != idx && tag1 != idx) {
          const int s = (tag0 == idx + 1) ? 1 : 0;
          m5_fft(wave, len, idx, n_frames, hop, L, cb, mag0 + 264 * s, false, lane);
          if (s) tag1 = idx; else tag0 = idx;
        }
        const int s0 = (tag0 == idx) ? 0 : 1;
        if ((s0 ? tag0 : tag1) != idx + 1) {
          m5_fft(wave, len, idx + 1, n_frames, hop, L, cb, mag0 + 264 * (1 - s0), false, lane);
          if (s0) tag0 = idx + 1; else tag1 = idx + 1;
        }
        const float* p0 = mag0 + 264 * s0;
        const float* p1 = mag0 + 264 * (1 - s0);
        float* pm = reinterpret_cast<float*>(cb);       // the interpolated power spectrum (the exchange buffer is free here)
        for (int k = lane; k < 257; k += 64) {
          const float v = alpha * p1[k] + (1.f - alpha) * p0[k];
          pm[k] = v * v;
        }
        m5_wave_sync();
        pw = pm;
      }
      if (lane < n_mels) {
        float s = 0.f;
        for (int i = 0; i < cntA; ++i) s = fmaf(wts[offA + i], pw[loA + i], s);
        melv[lane] = 10.f * log10f(fmaxf(s, 1e-10f));
      }
      for (int base = 64; base < n_mels; base += 16) {
        const int m = base + (lane >> 2), part = lane & 3;
        float s = 0.f;
        if (m < n_mels) {
          const int lo = range[2 * m], hi = range[2 * m + 1], off = sp_off[m];
          const int chunk = (hi - lo + 3) >> 2;
          const int k0 = lo + part * chunk, k1 = min(hi, k0 + chunk);
          for (int k = k0; k < k1; ++k) s = fmaf(wts[off + k - lo], pw[k], s);
        }
        s += __shfl_xor(s, 1, 64);
        s += __shfl_xor(s, 2, 64);
        if (m < n_mels && part == 0) melv[m] = 10.f * log10f(fmaxf(s, 1e-10f));
      }
      m5_wave_sync();
      float qs = 0.f;
      for (int m = lane; m < n_mels; m += 64) qs += melv[m] * melv[m];
      qs = wave_sum(qs);
      const float inv = 1.f / fmaxf(sqrtf(qs), 1e-12f);
      const bool tm = (tmask && tmask[(size_t)b * T_out + j]) || (j >= it0 && j < it1);
      for (int m = lane; m < n_mels; m += 64) {
        float v = melv[m] * inv;
        if (tm || (fmask && fmask[(size_t)b * n_mels + m]) || (m >= if0 && m < if1)) v = 0.f;
        tile[m * (MEL_FT + 1) + jj] = v;
      }
      m5_wave_sync();      // melv / the exchange buffer are rewritten by the next frame
    }
  } else {
    for (int i = tid; i < n_mels * MEL_FT; i += 256) tile[(i / MEL_FT) * (MEL_FT + 1) + i % MEL_FT] = 0.f;
  }
  __syncthreads();
  if (out) {
    for (int i = tid; i < n_mels * MEL_FT; i += 256) {
      const int m = i / MEL_FT, jj = i % MEL_FT;
      if (j0 + jj < T_out) out[((size_t)b * n_mels + m) * T_out + j0 + jj] = tile[m * (MEL_FT + 1) + jj];
    }
  }
  if (packed) {
    unsigned short* pb = packed + ((size_t)b * T_out + j0) * n_mels;
    const int nfr = min(MEL_FT, T_out - j0);
    for (int i = tid; i < nfr * n_mels; i += 256) {
      const int jj = i / n_mels, m = i - jj * n_mels;
      pb[i] = __builtin_bit_cast(unsigned short, (__bf16)tile[m * (MEL_FT + 1) + jj]);
    }
  }
}

extern "C" int tn_mel_create(int32_t sample_rate, int32_t n_fft, int32_t win_length, int32_t hop_length, int32_t n_mels,
                             tn_mel** out) {
  if (!out || n_fft < 16 || n_fft > 4096 || (n_fft & (n_fft - 1)) || win_length <= 0 || win_length > n_fft || hop_length <= 0 ||
      n_mels <= 0 || sample_rate <= 0)
    return TN_E_BADARG;
  tn_mel* m = new tn_mel();
  m->sample_rate = sample_rate; m->n_fft = n_fft; m->win_length = win_length; m->hop = hop_length; m->n_mels = n_mels;
  m->n_freqs = n_fft / 2 + 1;
  m->log2n = 0;
  while ((1 << m->log2n) < n_fft) ++m->log2n;
  const double PI = 3.14159265358979323846;
  std::vector<float> win(n_fft, 0.f), tw(n_fft), fb((size_t)n_mels * m->n_freqs, 0.f);
  std::vector<int> range(2 * n_mels);
  const int left = (n_fft - win_length) / 2;
  for (int n = 0; n < win_length; ++n) win[left + n] = (float)(0.5 - 0.5 * cos(2.0 * PI * n / win_length));
  for (int k = 0; k < n_fft / 2; ++k) { tw[2 * k] = (float)cos(2.0 * PI * k / n_fft); tw[2 * k + 1] = (float)(-sin(2.0 * PI * k / n_fft)); }
  // torchaudio.functional.melscale_fbanks(n_freqs, 0, sr/2, n_mels, sr, norm=None, mel_scale="htk")
  const double f_max = sample_rate / 2.0;
  const double m_min = 0.0, m_max = 2595.0 * log10(1.0 + f_max / 700.0);
  std::vector<double> f_pts(n_mels + 2);
  for (int i = 0; i < n_mels + 2; ++i) {
    const double mp = m_min + (m_max - m_min) * i / (n_mels + 1);
    f_pts[i] = 700.0 * (pow(10.0, mp / 2595.0) - 1.0);
  }
  for (int j = 0; j < n_mels; ++j) {
    int lo = m->n_freqs, hi = 0;
    for (int k = 0; k < m->n_freqs; ++k) {
      const double f = (double)(sample_rate / 2) * k / (m->n_freqs - 1);
      const double down = (f - f_pts[j]) / (f_pts[j + 1] - f_pts[j]);
      const double up = (f_pts[j + 2] - f) / (f_pts[j + 2] - f_pts[j + 1]);
      const double v = fmax(0.0, fmin(down, up));
      fb[(size_t)j * m->n_freqs + k] = (float)v;
      if (v > 0.0) { lo = k < lo ? k : lo; hi = k + 1; }
    }
    if (hi <= lo) { lo = 0; hi = 0; }
    range[2 * j] = lo; range[2 * j + 1] = hi;
  }
  TN_CHECK_HIP(hipMalloc(&m->window, win.size() * sizeof(float)));
  TN_CHECK_HIP(hipMalloc(&m->twiddle, tw.size() * sizeof(float)));
  TN_CHECK_HIP(hipMalloc(&m->fb, fb.size() * sizeof(float)));
  TN_CHECK_HIP(hipMalloc(&m->range, range.size() * sizeof(int)));
  TN_CHECK_HIP(hipMemcpy(m->window, win.data(), win.size() * sizeof(float), hipMemcpyHostToDevice));
  TN_CHECK_HIP(hipMemcpy(m->twiddle, tw.data(), tw.size() * sizeof(float), hipMemcpyHostToDevice));
  TN_CHECK_HIP(hipMemcpy(m->fb, fb.data(), fb.size() * sizeof(float), hipMemcpyHostToDevice));
  TN_CHECK_HIP(hipMemcpy(m->range, range.data(), range.size() * sizeof(int), hipMemcpyHostToDevice));
  if (n_fft == 512 && n_mels <= MEL5_MP) {
    std::vector<float> t512(8 * 64 * 2), t64(8 * 8 * 2), spw;
    std::vector<int> off(n_mels + 1, 0);
    for (int k1 = 0; k1 < 8; ++k1)
      for (int n2 = 0; n2 < 64; ++n2) {
        t512[2 * (k1 * 64 + n2)] = (float)cos(2.0 * PI * (k1 * n2) / 512.0);
        t512[2 * (k1 * 64 + n2) + 1] = (float)(-sin(2.0 * PI * (k1 * n2) / 512.0));
      }
    for (int j1 = 0; j1 < 8; ++j1)
      for (int m2 = 0; m2 < 8; ++m2) {
        t64[2 * (j1 * 8 + m2)] = (float)cos(2.0 * PI * (j1 * m2) / 64.0);
        t64[2 * (j1 * 8 + m2) + 1] = (float)(-sin(2.0 * PI * (j1 * m2) / 64.0));
      }
    for (int j = 0; j < n_mels; ++j) {
      off[j] = (int)spw.size();
      for (int k = range[2 * j]; k < range[2 * j + 1]; ++k) spw.push_back(fb[(size_t)j * m->n_freqs + k]);
    }
    off[n_mels] = (int)spw.size();
    m->nnz = (int)spw.size();
    if (m->nnz <= MEL5_MAXW) {
      if (spw.empty()) spw.push_back(0.f);
      TN_CHECK_HIP(hipMalloc(&m->tw512, t512.size() * sizeof(float)));
      TN_CHECK_HIP(hipMalloc(&m->tw64, t64.size() * sizeof(float)));
      TN_CHECK_HIP(hipMalloc(&m->sp_w, spw.size() * sizeof(float)));
      TN_CHECK_HIP(hipMalloc(&m->sp_off, off.size() * sizeof(int)));
      TN_CHECK_HIP(hipMemcpy(m->tw512, t512.data(), t512.size() * sizeof(float), hipMemcpyHostToDevice));
      TN_CHECK_HIP(hipMemcpy(m->tw64, t64.data(), t64.size() * sizeof(float), hipMemcpyHostToDevice));
      TN_CHECK_HIP(hipMemcpy(m->sp_w, spw.data(), spw.size() * sizeof(float), hipMemcpyHostToDevice));
      TN_CHECK_HIP(hipMemcpy(m->sp_off, off.data(), off.size() * sizeof(int), hipMemcpyHostToDevice));
      const char* e = getenv("TN_MEL_GENERIC");        // 1: the generic workgroup-per-frame kernel (A/B, tests)
      m->fast512 = !(e && atoi(e));
    }
  }
  *out = m;
  return 0;
}

extern "C" void tn_mel_destroy(tn_mel* m) {
  if (!m) return;
  (void)hipFree(m->window); (void)hipFree(m->twiddle); (void)hipFree(m->fb); (void)hipFree(m->range);
  (void)hipFree(m->tw512); (void)hipFree(m->tw64); (void)hipFree(m->sp_w); (void)hipFree(m->sp_off);
  delete m;
}

extern "C" int64_t tn_mel_num_frames(const tn_mel* m, int64_t n_samples) { return m ? 1 + n_samples / m->hop : 0; }

static int mel_batch_launch(tn_mel* m, const float* waves, int32_t batch, int64_t n_samples_max, const int64_t* lengths,
                            const double* rates, const uint8_t* freq_mask, const uint8_t* time_mask, int32_t frames_out,
                            float* out, unsigned short* packed, void* stream, const int32_t* imasks = nullptr);
extern "C" int tn_mel_forward_batch(tn_mel* m, const float* waves, int32_t batch, int64_t n_samples_max, const int64_t* lengths,
                                    const double* rates, const uint8_t* freq_mask, const uint8_t* time_mask, int32_t frames_out,
                                    float* out, void* stream) {
  if (!out) return TN_E_BADARG;
  return mel_batch_launch(m, waves, batch, n_samples_max, lengths, rates, freq_mask, time_mask, frames_out, out, nullptr, stream);
}
extern "C" int tn_mel_forward_batch_packed(tn_mel* m, const float* waves, int32_t batch, int64_t n_samples_max, const int64_t* lengths,
                                           const double* rates, const uint8_t* freq_mask, const uint8_t* time_mask,
                                           int32_t frames_out, void* packed_bf16, float* out_or_null, void* stream) {
  if (!packed_bf16) return TN_E_BADARG;
  return mel_batch_launch(m, waves, batch, n_samples_max, lengths, rates, freq_mask, time_mask, frames_out, out_or_null,
                          (unsigned short*)packed_bf16, stream);
}
static int mel_batch_launch(tn_mel* m, const float* waves, int32_t batch, int64_t n_samples_max, const int64_t* lengths,
                            const double* rates, const uint8_t* freq_mask, const uint8_t* time_mask, int32_t frames_out,
                            float* out, unsigned short* packed, void* stream, const int32_t* imasks) {
  if (!m || !waves || batch <= 0 || n_samples_max <= 0 || frames_out <= 0) return TN_E_BADARG;
  if (m->fast512) {
    const size_t smem5 = (size_t)4 * 8 * MEL5_CS * sizeof(float2) + (size_t)(4 * 2 * 264 + 4 * MEL5_MP + m->n_mels * (MEL_FT + 1) + m->nnz) * sizeof(float);
    hipLaunchKernelGGL(mel512_batch_kernel, dim3((frames_out + MEL_FT - 1) / MEL_FT, batch), dim3(256), smem5, (hipStream_t)stream, waves,
                       n_samples_max, lengths, rates, freq_mask, time_mask, frames_out, m->hop, m->n_mels, m->window,
                       reinterpret_cast<const float2*>(m->tw512), reinterpret_cast<const float2*>(m->tw64), m->sp_w, m->sp_off, m->range,
                       m->nnz, imasks, out, packed);
    return (int)hipGetLastError();
  }
  if (imasks) return TN_E_BADARG;
  const size_t smem = (size_t)(2 * m->n_fft + 2 * (m->n_freqs + 3) + m->n_mels + m->n_mels * (MEL_FT + 1)) * sizeof(float);
  const int threads = m->n_fft / 2 < 64 ? 64 : (m->n_fft / 2 > 1024 ? 1024 : m->n_fft / 2);
  hipLaunchKernelGGL(mel_batch_kernel, dim3((frames_out + MEL_FT - 1) / MEL_FT, batch), dim3(threads), smem, (hipStream_t)stream, waves,
                     n_samples_max, lengths, rates, freq_mask, time_mask, frames_out, m->n_fft, m->log2n, m->hop, m->n_mels, m->n_freqs,
                     m->window, m->fb, m->range, m->twiddle, out, packed);
  return (int)hipGetLastError();
}

extern "C" int tn_mel_forward(tn_mel* m, const float* waves, int32_t batch, int64_t n_samples, const int32_t* masks, float* out,
                              void* stream) {
  if (!m || !waves || !out || batch <= 0 || n_samples <= 0) return TN_E_BADARG;
  const int T = (int)(1 + n_samples / m->hop);
  if (m->fast512) return mel_batch_launch(m, waves, batch, n_samples, nullptr, nullptr, nullptr, nullptr, T, out, nullptr, stream, masks);
  const size_t smem = (size_t)(2 * m->n_fft + m->n_freqs + 3 + m->n_mels) * sizeof(float);
  const int threads = m->n_fft / 2 < 64 ? 64 : (m->n_fft / 2 > 1024 ? 1024 : m->n_fft / 2);
  hipLaunchKernelGGL(mel_frame_kernel, dim3(T, batch), dim3(threads), smem, (hipStream_t)stream, waves, n_samples, T, m->n_fft,
                     m->log2n, m->hop, m->n_mels, m->n_freqs, m->window, m->fb, m->range, m->twiddle, masks, out);
  return (int)hipGetLastError();
}
