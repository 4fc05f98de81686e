// Channels-last convolution plumbing for the build-defined ResNet-18 / 1-D CNN encoders (not in the
// reference): im2col gather, its adjoint (a deterministic gather over the windows covering a pixel --
// no atomics), and 2-D max pooling.  The convolutions themselves are msn_sgemm on the column matrix:
//   y[(b,oh,ow)][co] = sum_{(c,u,v)} cols[(b,oh,ow)][(c,u,v)] * W[co][(c,u,v)]
// with the column order (c, u, v) equal to the flattening of a (C_out, C_in, kh, kw) weight, so
// torchvision-style parameters are used as they are.  A 1-D convolution is the H = 1 case.
#include <algorithm>

#include "msn_common.h"

namespace msn {

struct ConvGeom {
    int B, H, W, C;          // input, channels-last (B, H, W, C)
    int kh, kw, sh, sw, ph, pw;
    int OH, OW;
};

__global__ void im2col_kernel(const float* __restrict__ x, ConvGeom g, float* __restrict__ cols) {
    const int64_t K = (int64_t)g.C * g.kh * g.kw;
    const int64_t total = (int64_t)g.B * g.OH * g.OW * K;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = i / K;
        const int kk = (int)(i % K);
        const int v = kk % g.kw, u = (kk / g.kw) % g.kh, c = kk / (g.kw * g.kh);
        const int ow = (int)(row % g.OW), oh = (int)((row / g.OW) % g.OH);
        const int64_t b = row / ((int64_t)g.OW * g.OH);
        const int y = oh * g.sh + u - g.ph, xx = ow * g.sw + v - g.pw;
        float val = 0.f;
        if (y >= 0 && y < g.H && xx >= 0 && xx < g.W) val = x[((b * g.H + y) * g.W + xx) * g.C + c];
        cols[i] = val;
    }
}
// dx[b,y,x,c] = sum over windows (oh, ow) and taps (u, v) with oh*sh + u - ph = y, ow*sw + v - pw = x
__global__ void col2im_kernel(const float* __restrict__ dcols, ConvGeom g, float* __restrict__ dx) {
    const int64_t K = (int64_t)g.C * g.kh * g.kw;
    const int64_t total = (int64_t)g.B * g.H * g.W * g.C;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % g.C);
        const int xx = (int)((i / g.C) % g.W), y = (int)((i / ((int64_t)g.C * g.W)) % g.H);
        const int64_t b = i / ((int64_t)g.C * g.W * g.H);
        float s = 0.f;
        for (int u = 0; u < g.kh; ++u) {
            const int ty = y + g.ph - u;
            if (ty < 0 || ty % g.sh != 0) continue;
            const int oh = ty / g.sh;
            if (oh >= g.OH) continue;
            for (int v = 0; v < g.kw; ++v) {
                const int tx = xx + g.pw - v;
                if (tx < 0 || tx % g.sw != 0) continue;
                const int ow = tx / g.sw;
                if (ow >= g.OW) continue;
                s += dcols[((b * g.OH + oh) * g.OW + ow) * K + ((int64_t)c * g.kh + u) * g.kw + v];
            }
        }
        dx[i] = s;
    }
}

__global__ void maxpool_fwd_kernel(const float* __restrict__ x, ConvGeom g, float* __restrict__ y, int* __restrict__ arg) {
    const int64_t total = (int64_t)g.B * g.OH * g.OW * g.C;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % g.C);
        const int ow = (int)((i / g.C) % g.OW), oh = (int)((i / ((int64_t)g.C * g.OW)) % g.OH);
        const int64_t b = i / ((int64_t)g.C * g.OW * g.OH);
        float best = -INFINITY;
        int bi = 0;
        for (int u = 0; u < g.kh; ++u) {
            const int yy = oh * g.sh + u - g.ph;
            if (yy < 0 || yy >= g.H) continue;
            for (int v = 0; v < g.kw; ++v) {
                const int xx = ow * g.sw + v - g.pw;
                if (xx < 0 || xx >= g.W) continue;
                const float val = x[((b * g.H + yy) * g.W + xx) * g.C + c];
                if (val > best) { best = val; bi = yy * g.W + xx; }
            }
        }
        y[i] = best;
        arg[i] = bi;
    }
}
// dx[b,y,x,c] = sum of dy over the windows whose arg-max is this pixel (gather: deterministic)
__global__ void maxpool_bwd_kernel(const float* __restrict__ dy, const int* __restrict__ arg, ConvGeom g,
                                   float* __restrict__ dx) {
    const int64_t total = (int64_t)g.B * g.H * g.W * g.C;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % g.C);
        const int xx = (int)((i / g.C) % g.W), y = (int)((i / ((int64_t)g.C * g.W)) % g.H);
        const int64_t b = i / ((int64_t)g.C * g.W * g.H);
        const int me = y * g.W + xx;
        float s = 0.f;
        for (int u = 0; u < g.kh; ++u) {
            const int ty = y + g.ph - u;
            if (ty < 0 || ty % g.sh != 0 || ty / g.sh >= g.OH) continue;
            for (int v = 0; v < g.kw; ++v) {
                const int tx = xx + g.pw - v;
                if (tx < 0 || tx % g.sw != 0 || tx / g.sw >= g.OW) continue;
                const int64_t o = ((b * g.OH + ty / g.sh) * g.OW + tx / g.sw) * g.C + c;
                if (arg[o] == me) s += dy[o];
            }
        }
        dx[i] = s;
    }
}


// 4 channels per thread (C % 4 == 0): 16-byte loads / stores of the pixel's channel group
__global__ void maxpool_fwd4_kernel(const float* __restrict__ x, ConvGeom g, float* __restrict__ y, int* __restrict__ arg) {
    const int C4 = g.C / 4;
    const int64_t total = (int64_t)g.B * g.OH * g.OW * C4;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c4 = (int)(i % C4);
        const int ow = (int)((i / C4) % g.OW), oh = (int)((i / ((int64_t)C4 * g.OW)) % g.OH);
        const int64_t b = i / ((int64_t)C4 * g.OW * g.OH);
        float best[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        int bi[4] = {0, 0, 0, 0};
        for (int u = 0; u < g.kh; ++u) {
            const int yy = oh * g.sh + u - g.ph;
            if (yy < 0 || yy >= g.H) continue;
            for (int v = 0; v < g.kw; ++v) {
                const int xx = ow * g.sw + v - g.pw;
                if (xx < 0 || xx >= g.W) continue;
                const float4 t = *reinterpret_cast<const float4*>(x + ((b * g.H + yy) * g.W + xx) * g.C + 4 * c4);
                const float val[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (val[k] > best[k]) { best[k] = val[k]; bi[k] = yy * g.W + xx; }
            }
        }
        reinterpret_cast<float4*>(y)[i] = make_float4(best[0], best[1], best[2], best[3]);
        reinterpret_cast<int4*>(arg)[i] = make_int4(bi[0], bi[1], bi[2], bi[3]);
    }
}
__global__ void maxpool_bwd4_kernel(const float* __restrict__ dy, const int* __restrict__ arg, ConvGeom g,
                                    float* __restrict__ dx) {
    const int C4 = g.C / 4;
    const int64_t total = (int64_t)g.B * g.H * g.W * C4;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c4 = (int)(i % C4);
        const int xx = (int)((i / C4) % g.W), y = (int)((i / ((int64_t)C4 * g.W)) % g.H);
        const int64_t b = i / ((int64_t)C4 * g.W * g.H);
        const int me = y * g.W + xx;
        float s[4] = {0.f, 0.f, 0.f, 0.f};
        for (int u = 0; u < g.kh; ++u) {
            const int ty = y + g.ph - u;
            if (ty < 0 || ty % g.sh != 0 || ty / g.sh >= g.OH) continue;
            for (int v = 0; v < g.kw; ++v) {
                const int tx = xx + g.pw - v;
                if (tx < 0 || tx % g.sw != 0 || tx / g.sw >= g.OW) continue;
                const int64_t o = ((b * g.OH + ty / g.sh) * g.OW + tx / g.sw) * g.C + 4 * c4;
                const int4 a = *reinterpret_cast<const int4*>(arg + o);
                const float4 d = *reinterpret_cast<const float4*>(dy + o);
                s[0] += a.x == me ? d.x : 0.f;
                s[1] += a.y == me ? d.y : 0.f;
                s[2] += a.z == me ? d.z : 0.f;
                s[3] += a.w == me ? d.w : 0.f;
            }
        }
        reinterpret_cast<float4*>(dx)[i] = make_float4(s[0], s[1], s[2], s[3]);
    }
}

// ---- tap-major columns: cols[(b,oh,ow)][(u,v,c)] -- channels fastest -------------------------------------------
// With the (c,u,v) order above neighbouring threads read neighbouring PIXELS of one channel (C * 4 bytes apart) and
// col2im reads its columns kh*kw floats apart: 2.7 TB/s on the ResNet stages.  With the channel fastest both kernels
// move whole 16-byte groups of channels of one pixel / one tap (C % 4 == 0; every conv but a 3-channel stem), and the
// weight is re-laid once per step to (C_out, kh, kw, C_in) -- a few MB.
__global__ void im2col_tap_kernel(const float* __restrict__ x, ConvGeom g, float* __restrict__ cols) {
    const int C4 = g.C / 4;
    const int64_t K4 = (int64_t)C4 * g.kh * g.kw;
    const int64_t total = (int64_t)g.B * g.OH * g.OW * K4;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = i / K4;
        const int kk = (int)(i % K4);
        const int c4 = kk % C4, tap = kk / C4, v = tap % g.kw, u = tap / g.kw;
        const int ow = (int)(row % g.OW), oh = (int)((row / g.OW) % g.OH);
        const int64_t b = row / ((int64_t)g.OW * g.OH);
        const int y = oh * g.sh + u - g.ph, xx = ow * g.sw + v - g.pw;
        float4 val = make_float4(0.f, 0.f, 0.f, 0.f);
        if (y >= 0 && y < g.H && xx >= 0 && xx < g.W)
            val = *reinterpret_cast<const float4*>(x + ((b * g.H + y) * g.W + xx) * g.C + 4 * c4);
        reinterpret_cast<float4*>(cols)[i] = val;
    }
}
__global__ void col2im_tap_kernel(const float* __restrict__ dcols, ConvGeom g, float* __restrict__ dx) {
    const int C4 = g.C / 4;
    const int64_t K = (int64_t)g.C * g.kh * g.kw;
    const int64_t total = (int64_t)g.B * g.H * g.W * C4;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c4 = (int)(i % C4);
        const int xx = (int)((i / C4) % g.W), y = (int)((i / ((int64_t)C4 * g.W)) % g.H);
        const int64_t b = i / ((int64_t)C4 * g.W * g.H);
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int u = 0; u < g.kh; ++u) {
            const int ty = y + g.ph - u;
            if (ty < 0 || ty % g.sh != 0) continue;
            const int oh = ty / g.sh;
            if (oh >= g.OH) continue;
            for (int v = 0; v < g.kw; ++v) {
                const int tx = xx + g.pw - v;
                if (tx < 0 || tx % g.sw != 0) continue;
                const int ow = tx / g.sw;
                if (ow >= g.OW) continue;
                const float4 t = *reinterpret_cast<const float4*>(dcols + ((b * g.OH + oh) * g.OW + ow) * K +
                                                                  (int64_t)(u * g.kw + v) * g.C + 4 * c4);
                s.x += t.x, s.y += t.y, s.z += t.z, s.w += t.w;
            }
        }
        reinterpret_cast<float4*>(dx)[i] = s;
    }
}
// (C_out, C_in, taps) <-> (C_out, taps, C_pad >= C_in); to_tap != 0: src is torch's layout, the channels C_in .. C_pad-1
// of the tap-major copy are zeros (a 3-channel stem runs as a 4-channel one); to_tap == 0 drops them again
__global__ void conv_weight_relayout_kernel(const float* __restrict__ src, int64_t co, int ci, int ci_pad, int taps,
                                            int to_tap, float* __restrict__ dst) {
    const int64_t total = co * (to_tap == 1 ? ci_pad : ci) * taps;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        if (to_tap == 1) {   // i indexes dst (o, tap, c) with c < ci_pad
            const int64_t o = i / ((int64_t)ci_pad * taps);
            const int rem = (int)(i % ((int64_t)ci_pad * taps));
            const int c = rem % ci_pad, t = rem / ci_pad;
            dst[i] = c < ci ? src[(o * ci + c) * taps + t] : 0.f;
        } else if (to_tap == 2) {   // i indexes dst (tap, o, c): the K-major operand of the implicit-GEMM dgrad (ci_pad == ci)
            const int c = (int)(i % ci);
            const int64_t o = (i / ci) % co, t = i / ((int64_t)ci * co);
            dst[i] = src[(o * ci + c) * taps + t];
        } else {        // i indexes dst (o, c, tap) with c < ci
            const int64_t o = i / ((int64_t)ci * taps);
            const int rem = (int)(i % ((int64_t)ci * taps));
            const int t = rem % taps, c = rem / taps;
            dst[i] = src[(o * taps + t) * ci_pad + c];
        }
    }
}
// rows x C -> rows x Cp (Cp >= C), zeros in the added channels
__global__ void pad_channels_kernel(const float* __restrict__ x, int64_t rows, int C, int Cp, float* __restrict__ out) {
    const int64_t total = rows * Cp;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % Cp);
        out[i] = c < C ? x[(i / Cp) * C + c] : 0.f;
    }
}

static int make_geom(const char* who, int B, int H, int W, int C, int kh, int kw, int sh, int sw, int ph, int pw,
                     ConvGeom* g) {
    MSN_REQUIRE(B > 0 && H > 0 && W > 0 && C > 0 && kh > 0 && kw > 0 && sh > 0 && sw > 0 && ph >= 0 && pw >= 0,
                "%s: bad geometry", who);
    const int OH = (H + 2 * ph - kh) / sh + 1, OW = (W + 2 * pw - kw) / sw + 1;
    MSN_REQUIRE(OH > 0 && OW > 0, "%s: kernel larger than the padded input", who);
    *g = ConvGeom{B, H, W, C, kh, kw, sh, sw, ph, pw, OH, OW};
    return MSN_OK;
}
static unsigned grid_for(int64_t total) { return (unsigned)std::min<int64_t>(cdiv(total, 256), 8192); }

}  // namespace msn

using namespace msn;

extern "C" int msn_im2col(const float* x, int B, int H, int W, int C, int kh, int kw, int sh, int sw, int ph, int pw,
                          float* cols, msn_stream_t stream) {
    ConvGeom g;
    if (int rc = make_geom("msn_im2col", B, H, W, C, kh, kw, sh, sw, ph, pw, &g)) return rc;
    MSN_REQUIRE(x && cols, "msn_im2col: null pointer");
    hipLaunchKernelGGL(im2col_kernel, dim3(grid_for((int64_t)B * g.OH * g.OW * C * kh * kw)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), x, g, cols);
    MSN_LAUNCH_CHECK();
    return MSN_OK;
}
extern "C" int msn_col2im(const float* dcols, int B, int H, int W, int C, int kh, int kw, int sh, int sw, int ph, int pw,
                          float* dx, msn_stream_t stream) {
    ConvGeom g;
    if (int rc = make_geom("msn_col2im", B, H, W, C, kh, kw, sh, sw, ph, pw, &g)) return rc;
    MSN_REQUIRE(dx && dcols, "msn_col2im: null pointer");
    hipLaunchKernelGGL(col2im_kernel, dim3(grid_for((int64_t)B * H * W * C)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       dcols, g, dx);
    MSN_LAUNCH_CHECK();
    return MSN_OK;
}
extern "C" int msn_im2col_tap(const float* x, int B, int H, int W, int C, int kh, int kw, int sh, int sw, int ph, int pw,
                              float* cols, msn_stream_t stream) {
    ConvGeom g;
    if (int rc = make_geom("msn_im2col_tap", B, H, W, C, kh, kw, sh, sw, ph, pw, &g)) return rc;
    MSN_REQUIRE(x && cols, "msn_im2col_tap: null pointer");
    MSN_REQUIRE(C % 4 == 0 && ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(cols)) & 15) == 0,
                "msn_im2col_tap: needs C %% 4 == 0 and 16-byte aligned tensors");
    hipLaunchKernelGGL(im2col_tap_kernel, dim3(grid_for((int64_t)B * g.OH * g.OW * (C / 4) * kh * kw)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), x, g, cols);
    MSN_LAUNCH_CHECK();
    return MSN_OK;
}
extern "C" int msn_col2im_tap(const float* dcols, int B, int H, int W, int C, int kh, int kw, int sh, int sw, int ph,
                              int pw, float* dx, msn_stream_t stream) {
    ConvGeom g;
    if (int rc = make_geom("msn_col2im_tap", B, H, W, C, kh, kw, sh, sw, ph, pw, &g)) return rc;
    MSN_REQUIRE(dx && dcols, "msn_col2im_tap: null pointer");
    MSN_REQUIRE(C % 4 == 0 && ((reinterpret_cast<uintptr_t>(dx) | reinterpret_cast<uintptr_t>(dcols)) & 15) == 0,
                "msn_col2im_tap: needs C %% 4 == 0 and 16-byte aligned tensors");
    hipLaunchKernelGGL(col2im_tap_kernel, dim3(grid_for((int64_t)B * H * W * (C / 4))), dim3(256), 0,
                       static_cast<hipStream_t>(stream), dcols, g, dx);
    MSN_LAUNCH_CHECK();
    return MSN_OK;
}
extern "C" int msn_conv_weight_relayout(const float* src, int64_t co, int ci, int ci_pad, int taps, int to_tap,
                                        float* dst, msn_stream_t stream) {
    MSN_REQUIRE(src && dst && co > 0 && ci > 0 && ci_pad >= ci && taps > 0, "msn_conv_weight_relayout: bad arguments");
    hipLaunchKernelGGL(conv_weight_relayout_kernel, dim3(grid_for(co * ci_pad * taps)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), src, co, ci, ci_pad, taps, to_tap, dst);
    MSN_LAUNCH_CHECK();
    return MSN_OK;
}
extern "C" int msn_pad_channels(const float* x, int64_t rows, int C, int Cp, float* out, msn_stream_t stream) {
    MSN_REQUIRE(x && out && rows > 0 && C > 0 && Cp >= C, "msn_pad_channels: bad arguments");
    hipLaunchKernelGGL(pad_channels_kernel, dim3(grid_for(rows * Cp)), dim3(256), 0, static_cast<hipStream_t>(stream), x,
                       rows, C, Cp, out);
    MSN_LAUNCH_CHECK();
    return MSN_OK;
}
extern "C" int msn_maxpool2d_fwd(const float* x, int B, int H, int W, int C, int k, int s, int p, float* y, int* argmax,
                                 msn_stream_t stream) {
    ConvGeom g;
    if (int rc = make_geom("msn_maxpool2d_fwd", B, H, W, C, k, k, s, s, p, p, &g)) return rc;
    MSN_REQUIRE(x && y && argmax, "msn_maxpool2d_fwd: null pointer");
    const bool vec = C % 4 == 0 && ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y) |
                                     reinterpret_cast<uintptr_t>(argmax)) & 15) == 0;
    if (vec)
        hipLaunchKernelGGL(maxpool_fwd4_kernel, dim3(grid_for((int64_t)B * g.OH * g.OW * (C / 4))), dim3(256), 0,
                           static_cast<hipStream_t>(stream), x, g, y, argmax);
    else
        hipLaunchKernelGGL(maxpool_fwd_kernel, dim3(grid_for((int64_t)B * g.OH * g.OW * C)), dim3(256), 0,
                           static_cast<hipStream_t>(stream), x, g, y, argmax);
    MSN_LAUNCH_CHECK();
    return MSN_OK;
}
extern "C" int msn_maxpool2d_bwd(const float* dy, const int* argmax, int B, int H, int W, int C, int k, int s, int p,
                                 float* dx, msn_stream_t stream) {
    ConvGeom g;
    if (int rc = make_geom("msn_maxpool2d_bwd", B, H, W, C, k, k, s, s, p, p, &g)) return rc;
    MSN_REQUIRE(dy && dx && argmax, "msn_maxpool2d_bwd: null pointer");
    const bool vec = C % 4 == 0 && ((reinterpret_cast<uintptr_t>(dy) | reinterpret_cast<uintptr_t>(dx) |
                                     reinterpret_cast<uintptr_t>(argmax)) & 15) == 0;
    if (vec)
        hipLaunchKernelGGL(maxpool_bwd4_kernel, dim3(grid_for((int64_t)B * H * W * (C / 4))), dim3(256), 0,
                           static_cast<hipStream_t>(stream), dy, argmax, g, dx);
    else
        hipLaunchKernelGGL(maxpool_bwd_kernel, dim3(grid_for((int64_t)B * H * W * C)), dim3(256), 0,
                           static_cast<hipStream_t>(stream), dy, argmax, g, dx);
    MSN_LAUNCH_CHECK();
    return MSN_OK;
}
