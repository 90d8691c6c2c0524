// Generic (any layer widths <= HPV_MAXH, any variational form) fp64 kernels of the hp-VPINN
// iteration for gfx950.  One thread per quadrature point for the network, one workgroup per
// element for the projection.  These are the shape-agnostic path and the on-device cross-check
// of the MFMA path (kernels_mfma.hip), which takes over for the 20-wide BASELINE networks.
//
// Maths (replaces tf.gradients at P1:144-148, P2:175-185, P3:236-245):
//   hidden layer:  z = hW+b, z_c = h_c W, z_cc = h_cc W
//                  h' = s(z), h'_c = s'(z) z_c, h'_cc = s''(z) z_c^2 + s'(z) z_cc
//   reverse:       zb_cc = hb'_cc s',  zb_c = hb'_c s' + 2 hb'_cc s'' z_c,
//                  zb   = hb' s' + sum_c [hb'_c s'' z_c + hb'_cc (s''' z_c^2 + s'' z_cc)]
#include "hpv_internal.h"
#include "hpv_math.h"

#define WAVE 64

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, WAVE);
    return v;
}

// Block-wide sum of one double per thread (blockDim.x multiple of 64, <= 1024); result valid in thread 0.
__device__ __forceinline__ double block_sum(double v, double* scratch) {
    v = wave_sum(v);
    int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) scratch[w] = v;
    __syncthreads();
    double r = 0.0;
    if (threadIdx.x == 0) {
        int nw = (blockDim.x + 63) >> 6;
        for (int i = 0; i < nw; ++i) r += scratch[i];
    }
    return r;
}

__device__ __forceinline__ void act_eval(int act, double z, double& a, double& a1) {
    if (act == HPV_ACT_TANH) {
        a = hpv_tanh(z);
        a1 = 1.0 - a * a;
    } else {
        hpv_sincos(z, &a, &a1);
    }
}
// s'' and s''' from the saved (s, s').
__device__ __forceinline__ void act_hi(int act, double a, double a1, double& a2, double& a3) {
    if (act == HPV_ACT_TANH) {
        a2 = -2.0 * a * a1;
        a3 = -2.0 * a1 * (1.0 - 3.0 * a * a);
    } else {
        a2 = -a;
        a3 = -a1;
    }
}

// ------------------------------------------------------------------------------------------------
// Forward Taylor-mode MLP: value + tangent channels at every point.  (P1:128-148)
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_mlp_fwd_generic(NetDesc nd, const double* __restrict__ theta,
                                                        const double* __restrict__ X, double* __restrict__ ACT,
                                                        double* __restrict__ OUT, long N, int save_act) {
    long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= N) return;
    const int C = nd.C, nT1 = nd.nT1, nT2 = nd.nT2;
    const bool mixed = nT1 == 2 && nT2 == 1;      // the mixed second tangent (NetDesc::t2w)
    double h[HPV_MAXC][HPV_MAXH];
    double zn[HPV_MAXC][HPV_MAXH];
    for (int j = 0; j < nd.d; ++j) {
        h[0][j] = X[(long)j * N + p];
        for (int a = 0; a < nT1; ++a) h[1 + a][j] = (nd.t1dim[a] == j) ? 1.0 : 0.0;
        for (int b = 0; b < nT2; ++b) h[1 + nT1 + b][j] = 0.0;
    }
    const int nhid = nd.nl - 1;
    for (int l = 0; l < nhid; ++l) {
        const int in = nd.width[l], out = nd.width[l + 1];
        const double* W = theta + nd.woff[l];
        const double* B = theta + nd.boff[l];
        for (int k = 0; k < out; ++k) {
            for (int ch = 0; ch < C; ++ch) {
                double acc = (ch == 0) ? B[k] : 0.0;
                for (int j = 0; j < in; ++j) acc += h[ch][j] * W[j * out + k];
                zn[ch][k] = acc;
            }
        }
        double* base = ACT + nd.actoff[l] * N;
        for (int k = 0; k < out; ++k) {
            double a, a1, a2, a3;
            act_eval(nd.act, zn[0][k], a, a1);
            act_hi(nd.act, a, a1, a2, a3);
            if (save_act) {
                base[((long)0 * out + k) * N + p] = a;
                base[((long)1 * out + k) * N + p] = a1;
            }
            h[0][k] = a;
            for (int t = 0; t < nT1; ++t) {
                double zc = zn[1 + t][k];
                if (save_act) base[((long)(2 + t) * out + k) * N + p] = zc;
                h[1 + t][k] = a1 * zc;
            }
            for (int b = 0; b < nT2; ++b) {
                double zc = zn[1 + nd.t2idx[b]][k];
                double zcc = zn[1 + nT1 + b][k];
                if (save_act) base[((long)(2 + nT1 + b) * out + k) * N + p] = zcc;
                if (mixed) h[1 + nT1 + b][k] = a2 * fma(nd.t2w[0], zn[1][k] * zn[1][k], nd.t2w[1] * (zn[2][k] * zn[2][k])) + a1 * zcc;
                else h[1 + nT1 + b][k] = a2 * zc * zc + a1 * zcc;
            }
        }
    }
    {   // linear head (P1:135-137)
        const int in = nd.width[nd.nl - 1];
        const double* W = theta + nd.woff[nd.nl - 1];
        const double* B = theta + nd.boff[nd.nl - 1];
        for (int ch = 0; ch < C; ++ch) {
            double acc = (ch == 0) ? B[0] : 0.0;
            for (int j = 0; j < in; ++j) acc += h[ch][j] * W[j];
            OUT[(long)ch * N + p] = acc;
        }
    }
}

void launch_mlp_fwd_generic(const NetDesc& nd, const double* theta, const double* X, double* ACT, double* OUT, long N,
                            int save_act, hipStream_t s) {
    if (N <= 0) return;
    dim3 grid((unsigned)((N + 255) / 256));
    hipLaunchKernelGGL(k_mlp_fwd_generic, grid, dim3(256), 0, s, nd, theta, X, ACT, OUT, N, save_act);
}

// ------------------------------------------------------------------------------------------------
// Reverse pass through the Taylor-mode forward: dL/dW, dL/db from the output adjoints GBAR.
// Each wave owns one row of GPART (zeroed by the host before launch); per parameter the wave
// reduces its 64 points with shuffles and lane 0 accumulates -- deterministic, no atomics.
// ------------------------------------------------------------------------------------------------
#define BWD_BLOCK 256
#define BWD_MAX_BLOCKS 1024

int mlp_bwd_generic_rows(long N) {
    long blocks = (N + BWD_BLOCK - 1) / BWD_BLOCK;
    if (blocks > BWD_MAX_BLOCKS) blocks = BWD_MAX_BLOCKS;
    if (blocks < 1) blocks = 1;
    return (int)(blocks * (BWD_BLOCK / WAVE));
}

__device__ __forceinline__ void load_layer_outputs(const NetDesc& nd, const double* ACT, int l, long N, long p, bool valid,
                                                   double hin[HPV_MAXC][HPV_MAXH]) {
    // outputs (h, h_c, h_cc) of hidden layer l recomputed from its saved slots
    const int w = nd.width[l + 1], nT1 = nd.nT1, nT2 = nd.nT2;
    const double* base = ACT + nd.actoff[l] * N;
    for (int j = 0; j < w; ++j) {
        double a = valid ? base[((long)0 * w + j) * N + p] : 0.0;
        double a1 = valid ? base[((long)1 * w + j) * N + p] : 0.0;
        double a2, a3;
        act_hi(nd.act, a, a1, a2, a3);
        hin[0][j] = a;
        for (int t = 0; t < nT1; ++t) {
            double zc = valid ? base[((long)(2 + t) * w + j) * N + p] : 0.0;
            hin[1 + t][j] = a1 * zc;
        }
        for (int b = 0; b < nT2; ++b) {
            double zc = valid ? base[((long)(2 + nd.t2idx[b]) * w + j) * N + p] : 0.0;
            double zcc = valid ? base[((long)(2 + nT1 + b) * w + j) * N + p] : 0.0;
            if (nT1 == 2 && nT2 == 1) {      // the mixed second tangent (NetDesc::t2w)
                const double z0 = valid ? base[((long)2 * w + j) * N + p] : 0.0, z1 = valid ? base[((long)3 * w + j) * N + p] : 0.0;
                hin[1 + nT1 + b][j] = a2 * fma(nd.t2w[0], z0 * z0, nd.t2w[1] * (z1 * z1)) + a1 * zcc;
            } else {
                hin[1 + nT1 + b][j] = a2 * zc * zc + a1 * zcc;
            }
        }
    }
}

__global__ void __launch_bounds__(BWD_BLOCK) k_mlp_bwd_generic(NetDesc nd, const double* __restrict__ theta,
                                                              const double* __restrict__ X,
                                                              const double* __restrict__ ACT,
                                                              const double* __restrict__ GBAR,
                                                              double* __restrict__ GPART, long N) {
    const int lane = threadIdx.x & 63;
    const long tid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long nthreads = (long)gridDim.x * blockDim.x;
    const long wave_global = tid >> 6;
    double* row = GPART + wave_global * (long)nd.P;
    const int C = nd.C, nT1 = nd.nT1, nT2 = nd.nT2;
    const int nhid = nd.nl - 1;

    double hbar[HPV_MAXC][HPV_MAXH];
    double zbar[HPV_MAXC][HPV_MAXH];
    double hin[HPV_MAXC][HPV_MAXH];

    for (long p0 = wave_global * WAVE; p0 < N; p0 += nthreads) {
        const long p = p0 + lane;
        const bool valid = p < N;
        double gb[HPV_MAXC];
        for (int ch = 0; ch < C; ++ch) gb[ch] = valid ? GBAR[(long)ch * N + p] : 0.0;

        // ---- linear head ----
        {
            const int in = nd.width[nd.nl - 1];
            const double* W = theta + nd.woff[nd.nl - 1];
            if (nhid > 0) {
                load_layer_outputs(nd, ACT, nhid - 1, N, p, valid, hin);
            } else {
                for (int j = 0; j < in; ++j) {
                    hin[0][j] = valid ? X[(long)j * N + p] : 0.0;
                    for (int t = 0; t < nT1; ++t) hin[1 + t][j] = (valid && nd.t1dim[t] == j) ? 1.0 : 0.0;
                    for (int b = 0; b < nT2; ++b) hin[1 + nT1 + b][j] = 0.0;
                }
            }
            for (int j = 0; j < in; ++j) {
                double v = 0.0;
                for (int ch = 0; ch < C; ++ch) {
                    v += hin[ch][j] * gb[ch];
                    hbar[ch][j] = gb[ch] * W[j];
                }
                v = wave_sum(v);
                if (lane == 0) row[nd.woff[nd.nl - 1] + j] += v;
            }
            double v = wave_sum(gb[0]);
            if (lane == 0) row[nd.boff[nd.nl - 1]] += v;
        }

        // ---- hidden layers, last to first ----
        for (int l = nhid - 1; l >= 0; --l) {
            const int in = nd.width[l], out = nd.width[l + 1];
            const double* W = theta + nd.woff[l];
            const double* base = ACT + nd.actoff[l] * N;
            for (int k = 0; k < out; ++k) {
                double a = valid ? base[((long)0 * out + k) * N + p] : 0.0;
                double a1 = valid ? base[((long)1 * out + k) * N + p] : 0.0;
                double a2, a3;
                act_hi(nd.act, a, a1, a2, a3);
                double zc[2] = {0.0, 0.0};
                for (int t = 0; t < nT1; ++t) zc[t] = valid ? base[((long)(2 + t) * out + k) * N + p] : 0.0;
                double zb = hbar[0][k] * a1;
                for (int t = 0; t < nT1; ++t) {
                    zbar[1 + t][k] = hbar[1 + t][k] * a1;
                    zb += hbar[1 + t][k] * a2 * zc[t];
                }
                for (int b = 0; b < nT2; ++b) {
                    const int t = nd.t2idx[b];
                    double zcc = valid ? base[((long)(2 + nT1 + b) * out + k) * N + p] : 0.0;
                    double hb = hbar[1 + nT1 + b][k];
                    zbar[1 + nT1 + b][k] = hb * a1;
                    if (nT1 == 2 && nT2 == 1) {      // the mixed second tangent (NetDesc::t2w) rides on both first tangents
                        zbar[1][k] += 2.0 * hb * a2 * nd.t2w[0] * zc[0];
                        zbar[2][k] += 2.0 * hb * a2 * nd.t2w[1] * zc[1];
                        zb += hb * (a3 * fma(nd.t2w[0], zc[0] * zc[0], nd.t2w[1] * (zc[1] * zc[1])) + a2 * zcc);
                    } else {
                        zbar[1 + t][k] += 2.0 * hb * a2 * zc[t];
                        zb += hb * (a3 * zc[t] * zc[t] + a2 * zcc);
                    }
                }
                zbar[0][k] = valid ? zb : 0.0;
            }
            // inputs of this layer
            if (l > 0) {
                load_layer_outputs(nd, ACT, l - 1, N, p, valid, hin);
            } else {
                for (int j = 0; j < in; ++j) {
                    hin[0][j] = valid ? X[(long)j * N + p] : 0.0;
                    for (int t = 0; t < nT1; ++t) hin[1 + t][j] = (valid && nd.t1dim[t] == j) ? 1.0 : 0.0;
                    for (int b = 0; b < nT2; ++b) hin[1 + nT1 + b][j] = 0.0;
                }
            }
            for (int j = 0; j < in; ++j) {
                for (int k = 0; k < out; ++k) {
                    double v = 0.0;
                    for (int ch = 0; ch < C; ++ch) v += hin[ch][j] * zbar[ch][k];
                    v = wave_sum(v);
                    if (lane == 0) row[nd.woff[l] + j * out + k] += v;
                }
            }
            for (int k = 0; k < out; ++k) {
                double v = wave_sum(zbar[0][k]);
                if (lane == 0) row[nd.boff[l] + k] += v;
            }
            if (l > 0) {
                for (int j = 0; j < in; ++j) {
                    for (int ch = 0; ch < C; ++ch) {
                        double acc = 0.0;
                        for (int k = 0; k < out; ++k) acc += zbar[ch][k] * W[j * out + k];
                        hin[ch][j] = acc;  // reuse hin as the new hbar
                    }
                }
                for (int j = 0; j < in; ++j)
                    for (int ch = 0; ch < C; ++ch) hbar[ch][j] = hin[ch][j];
            }
        }
    }
}

void launch_mlp_bwd_generic(const NetDesc& nd, const double* theta, const double* X, const double* ACT,
                            const double* GBAR, double* GPART, int rows, long N, hipStream_t s) {
    if (N <= 0) return;
    (void)hipMemsetAsync(GPART, 0, (size_t)rows * nd.P * sizeof(double), s);
    int blocks = rows / (BWD_BLOCK / WAVE);
    hipLaunchKernelGGL(k_mlp_bwd_generic, dim3(blocks), dim3(BWD_BLOCK), 0, s, nd, theta, X, ACT, GBAR, GPART, N);
}

// ------------------------------------------------------------------------------------------------
// Per-element projection onto the test functions, residual, element loss and (optionally) the
// adjoint back to the integrand channels.  One workgroup per element, sum-factorised
// (x-contraction then y-contraction).  Replaces the Python-unrolled reduce chains of
// P1:82-96, P2:91-120, P3:157-182.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_project(ProjDesc pd, const double* __restrict__ OUT, double* __restrict__ GBAR,
                                                double* __restrict__ R, const double* __restrict__ F,
                                                const double* __restrict__ coef, long coef_stride,
                                                const double* __restrict__ wtx, const double* __restrict__ wty,
                                                const double* __restrict__ eps_ptr, double* __restrict__ loss_e,
                                                double* __restrict__ deps_e, long N, int do_adjoint,
                                                const double* __restrict__ edge_u, const double* __restrict__ edge_dphi,
                                                const double* __restrict__ edge_coef, double* __restrict__ edge_gbar) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    const long e = blockIdx.x;
    const int qx = pd.qx, qy = pd.qy, ntx = pd.ntx, nty = pd.nty;
    const int NQ = qx * qy, NR = ntx * nty, nterms = pd.nterms, C = pd.C;
    double* G = sm;
    double* T = G + NQ;
    double* U = T + qy * ntx;
    double* Uraw = U + NR;
    double* S = Uraw + nterms * NR;
    double* red = S + nterms * nty * qx;
    const long base = e * NQ;
    const double eps = eps_ptr ? eps_ptr[0] : 0.0;
    const int tid = threadIdx.x, nthr = blockDim.x;

    for (int idx = tid; idx < NR; idx += nthr) U[idx] = F ? -F[e * NR + idx] : 0.0;
    __syncthreads();

    for (int t = 0; t < nterms; ++t) {
        const TermDesc& td = pd.t[t];
        double alpha[HPV_MAXC];
        for (int ch = 0; ch < C; ++ch) alpha[ch] = td.a0[ch] + eps * td.a1[ch];
        for (int q = tid; q < NQ; q += nthr) {
            double g = 0.0;
            for (int ch = 0; ch < C; ++ch)
                if (alpha[ch] != 0.0) g += alpha[ch] * OUT[(long)ch * N + base + q];
            G[q] = g;
        }
        __syncthreads();
        const double* ax = wtx + (long)td.dx * ntx * qx;
        for (int idx = tid; idx < qy * ntx; idx += nthr) {
            const int j = idx / ntx, r = idx - j * ntx;
            double acc = 0.0;
            for (int i = 0; i < qx; ++i) acc += ax[r * qx + i] * G[j * qx + i];
            T[idx] = acc;
        }
        __syncthreads();
        const double* by = wty + (long)td.dy * nty * qy;
        const double c = coef[(long)t * coef_stride + e];
        for (int idx = tid; idx < NR; idx += nthr) {
            const int k = idx / ntx, r = idx - k * ntx;
            double acc = 0.0;
            for (int j = 0; j < qy; ++j) acc += by[k * qy + j] * T[j * ntx + r];
            acc *= c;
            Uraw[t * NR + idx] = acc;
            U[idx] += td.eps_mult ? eps * acc : acc;
        }
        __syncthreads();
    }
    if (pd.edge) {  // P1:90: + 1/J [u(x_R) phi'_k(1) - u(x_L) phi'_k(-1)]
        const double uL = edge_u[2 * e], uR = edge_u[2 * e + 1], ce = edge_coef[e];
        for (int idx = tid; idx < NR; idx += nthr)
            U[idx] += ce * (uR * edge_dphi[2 * idx + 1] - uL * edge_dphi[2 * idx]);
        __syncthreads();
    }
    const int nax = pd.nact ? pd.nact[e] : ntx;          // active test functions of this element (P1:67)
    const double NRa = (double)(nax * nty);
    double sq = 0.0;
    for (int idx = tid; idx < NR; idx += nthr) {
        const double u = (idx % ntx) < nax ? U[idx] : 0.0;
        U[idx] = u;
        R[e * NR + idx] = u;
        sq += u * u;
    }
    sq = block_sum(sq, red);
    if (tid == 0) loss_e[e] = sq / NRa;  // reduce_mean(square(Res)) (P1:95)
    if (!do_adjoint) return;
    __syncthreads();

    const double sc = 2.0 / NRa;
    for (int t = 0; t < nterms; ++t) {
        const double* ax = wtx + (long)pd.t[t].dx * ntx * qx;
        for (int idx = tid; idx < nty * qx; idx += nthr) {
            const int k = idx / qx, i = idx - k * qx;
            double acc = 0.0;
            for (int r = 0; r < ntx; ++r) acc += ax[r * qx + i] * U[k * ntx + r];
            S[t * nty * qx + idx] = acc * sc;
        }
    }
    __syncthreads();
    double deps = 0.0;
    for (int q = tid; q < NQ; q += nthr) {
        const int j = q / qx, i = q - j * qx;
        double gb[HPV_MAXC];
        double o[HPV_MAXC];
        for (int ch = 0; ch < C; ++ch) {
            gb[ch] = 0.0;
            o[ch] = OUT[(long)ch * N + base + q];
        }
        for (int t = 0; t < nterms; ++t) {
            const TermDesc& td = pd.t[t];
            const double* by = wty + (long)td.dy * nty * qy;
            double gh = 0.0;
            for (int k = 0; k < nty; ++k) gh += by[k * qy + j] * S[t * nty * qx + k * qx + i];
            gh *= coef[(long)t * coef_stride + e];
            const double m = td.eps_mult ? eps : 1.0;
            double g1 = 0.0, gt = 0.0;
            for (int ch = 0; ch < C; ++ch) {
                const double al = td.a0[ch] + eps * td.a1[ch];
                gb[ch] += al * (m * gh);
                g1 += td.a1[ch] * o[ch];
                gt += al * o[ch];
            }
            deps += gh * (m * g1 + (td.eps_mult ? gt : 0.0));
        }
        for (int ch = 0; ch < C; ++ch) GBAR[(long)ch * N + base + q] = gb[ch];
    }
    if (pd.edge) {
        double sl = 0.0, sr = 0.0;
        for (int idx = tid; idx < NR; idx += nthr) {
            sl += U[idx] * edge_dphi[2 * idx];
            sr += U[idx] * edge_dphi[2 * idx + 1];
        }
        sl = block_sum(sl, red);
        sr = block_sum(sr, red);
        if (tid == 0) {
            edge_gbar[2 * e] = -edge_coef[e] * sc * sl;
            edge_gbar[2 * e + 1] = edge_coef[e] * sc * sr;
        }
    }
    if (pd.has_eps) {
        deps = block_sum(deps, red);
        if (tid == 0) deps_e[e] = deps;
    }
}

void launch_project(const ProjDesc& pd, const double* OUT, double* GBAR, double* R, const double* F, const double* coef,
                    long coef_stride, const double* wtx, const double* wty, const double* eps_ptr, double* loss_e,
                    double* deps_e, long N, long n_elem, int do_adjoint, const double* edge_u, const double* edge_dphi,
                    const double* edge_coef, double* edge_gbar, hipStream_t s) {
    if (n_elem <= 0) return;
    size_t lds = hpv_proj_lds_bytes(pd);
    hipLaunchKernelGGL(k_project, dim3((unsigned)n_elem), dim3(256), lds, s, pd, OUT, GBAR, R, F, coef, coef_stride, wtx,
                       wty, eps_ptr, loss_e, deps_e, N, do_adjoint, edge_u, edge_dphi, edge_coef, edge_gbar);
}

// ------------------------------------------------------------------------------------------------
// Boundary / data term: lossb = w * mean((u_d - u)^2)  (P1:98, P2:122, P3:184)
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_data_loss(const double* __restrict__ U, const double* __restrict__ Ud,
                                                  double* __restrict__ GBAR, double scale_grad,
                                                  double* __restrict__ part, int n) {
    __shared__ double red[16];
    double sq = 0.0;
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < n; p += gridDim.x * blockDim.x) {
        const double d = Ud[p] - U[p];
        sq += d * d;
        if (GBAR) GBAR[p] = scale_grad * d;
    }
    sq = block_sum(sq, red);
    if (threadIdx.x == 0) part[blockIdx.x] = sq;
}

void launch_data_loss(const double* U, const double* Ud, double* GBAR, double scale_grad, double* part, int n,
                      hipStream_t s) {
    if (n <= 0) return;
    int blocks = (n + 255) / 256;
    if (blocks > 64) blocks = 64;
    hipLaunchKernelGGL(k_data_loss, dim3(blocks), dim3(256), 0, s, U, Ud, GBAR, scale_grad, part, n);
}

// ------------------------------------------------------------------------------------------------
// Finalize: fixed-order sums of all partials into the packed reduce buffer
//   RB = [grad (P) | (d eps) | lossv | w*lossb | mean-square of the data term | pad]
// ------------------------------------------------------------------------------------------------
// TF1 AdamOptimizer rule (tf.train.AdamOptimizer(LR).minimize, P1:103-104), one parameter:
//   lr_t = lr sqrt(1-b2^t)/(1-b1^t); m = b1 m + (1-b1) g; v = b2 v + (1-b2) g^2; theta -= lr_t m/(sqrt(v)+eps)
// (eps OUTSIDE the bias correction, unlike torch.optim.Adam).  b1^t, b2^t are running products like TF's
// beta*_power variables; they are kept REPLICATED -- state[0..1] for the scalar block / k_adam and one copy per
// gradient block of the fused finalize+Adam kernel -- so that no block reads a value another block updates.
__device__ __forceinline__ void adam_update(const AdamArgs& ad, int i, double g, double b1p, double b2p) {
    double mi, vi, ti;
    hpv_adam_one(ad.lr, ad.b1, ad.b2, ad.eps, b1p, b2p, g, ad.m[i], ad.v[i], ad.theta[i], mi, vi, ti);
    ad.m[i] = mi;
    ad.v[i] = vi;
    ad.theta[i] = ti;
}

#define FIN_COLS 16                          // parameters per block (one 128-B row segment per partial row)
// FIN_THREADS / FIN_COLS row groups are summed in parallel: 16 (a 256-thread block: 4 waves to launch instead of 16, eight
// row loads in flight per thread) up to 1 024 rows, 64 beyond -- the kernel is a pure latency chain whose floor (kernarg
// fetch, one load round trip, the store drain: ~4.7 us behind another kernel) is most of its duration.  Config 4 (256 rows):
// 66.8 -> 65.9 us per iteration with the small block
template <int FIN_THREADS>
__global__ void __launch_bounds__(FIN_THREADS) k_finalize(const double* __restrict__ GPART_v, int rows_v,
                                                         const double* __restrict__ GPART_b, int rows_b,
                                                         const double* __restrict__ GPART_e, int rows_e,
                                                         const double* __restrict__ loss_e, long n_elem,
                                                         const double* __restrict__ deps_e,
                                                         const double* __restrict__ data_part, int n_data_part,
                                                         double lossb_weight, int n_data, int P, int has_eps,
                                                         double* RB, int write_grad, AdamArgs ad,
                                                         const int* __restrict__ xerr, unsigned int* __restrict__ xiter_bump, int pend) {
    // pend (the multi-GPU iteration in two launches, round 5): `ad` is the update of the PREVIOUS iteration -- its all-reduced
    // gradient is still in RB, the iteration kernel in front of this launch has already computed with the updated parameters
    // (k_iter_fused prologue, same arithmetic: hpv_adam_one) -- applied here, before RB takes this iteration's partial sums.  The
    // failure flag of the reduced buffer was latched into *xerr by that prologue.
    constexpr int FIN_PARTS = FIN_THREADS / FIN_COLS;
    __shared__ double red[FIN_PARTS * FIN_COLS];
    const int Ptot = P + (has_eps ? 1 : 0);
    if (blockIdx.x < gridDim.x - 1) {
        // 16 parameters per block; 64 row-groups summed in parallel (few dependent loads per thread: the kernel is
        // pure latency), then combined in a fixed order
        if (!write_grad) return;
        const int c = threadIdx.x & (FIN_COLS - 1), part = threadIdx.x / FIN_COLS;
        const int idx = blockIdx.x * FIN_COLS + c;
        // the Adam operands of this block's 16 parameters are requested BEFORE the row sums, so that their memory round trip
        // overlaps the rows' instead of following it (the kernel is one latency chain)
        bool upd = ad.theta && part == 0 && idx < P;
        double m0 = 0.0, v0 = 0.0, th0 = 0.0, b1p = 0.0, b2p = 0.0, gold = 0.0;
        int failed = 0;
        if (upd) {
            m0 = ad.m[idx]; v0 = ad.v[idx]; th0 = ad.theta[idx];
            b1p = ad.state[2 * (blockIdx.x + 1)]; b2p = ad.state[2 * (blockIdx.x + 1) + 1];
            if (pend) gold = RB[idx];      // the previous iteration's reduced gradient (this thread overwrites the slot below)
            // a SPLIT-mode barrier of this (or an earlier) iteration failed: no update.  pend: the update belongs to the PREVIOUS iteration --
            // the verdict the prologue of the iteration kernel formed from the reduced buffer (identical on every rank)
            if (xerr) failed = pend ? xerr[1] : *xerr;
        }
        double acc = 0.0;
        if (idx < P) {
            if (GPART_v) {
                // FIN_U independent loads in flight per thread (the sums stay in row order: same result as the plain loop)
#ifndef HPV_FIN_U
#define HPV_FIN_U 8
#endif
                constexpr int FIN_U = HPV_FIN_U;
                int r = part;
                for (; r + (FIN_U - 1) * FIN_PARTS < rows_v; r += FIN_U * FIN_PARTS) {
                    double t[FIN_U];
#pragma unroll
                    for (int u = 0; u < FIN_U; ++u) t[u] = GPART_v[(long)(r + u * FIN_PARTS) * P + idx];
#pragma unroll
                    for (int u = 0; u < FIN_U; ++u) acc += t[u];
                }
                for (; r < rows_v; r += FIN_PARTS) acc += GPART_v[(long)r * P + idx];
            }
            if (GPART_b) for (int r = part; r < rows_b; r += FIN_PARTS) acc += GPART_b[(long)r * P + idx];
            if (GPART_e) for (int r = part; r < rows_e; r += FIN_PARTS) acc += GPART_e[(long)r * P + idx];
        }
        red[part * FIN_COLS + c] = acc;
        __syncthreads();
        if (part == 0 && idx < P) {
            double t = 0.0;
#pragma unroll 8
            for (int k = 0; k < FIN_PARTS; ++k) t += red[k * FIN_COLS + c];
            RB[idx] = t;
            if (upd && !failed && pend) {      // the deferred update of the previous iteration
                double mi, vi, ti;
                hpv_adam_one(ad.lr, ad.b1, ad.b2, ad.eps, b1p, b2p, gold, m0, v0, th0, mi, vi, ti);
                ad.m[idx] = mi; ad.v[idx] = vi; ad.theta[idx] = ti;
                if (threadIdx.x == 0) {
                    ad.state[2 * (blockIdx.x + 1)] = b1p * ad.b1;
                    ad.state[2 * (blockIdx.x + 1) + 1] = b2p * ad.b2;
                }
            } else if (upd && !failed) {   // adam_update with the operands fetched above (same arithmetic, same order)
                double mi, vi, ti;
                hpv_adam_one(ad.lr, ad.b1, ad.b2, ad.eps, b1p, b2p, t, m0, v0, th0, mi, vi, ti);
                ad.m[idx] = mi;
                ad.v[idx] = vi;
                ad.theta[idx] = ti;
                // this block's private copy of the running beta powers (no cross-block race); its only readers are the 16
                // `upd` lanes of this wave, which consumed them above -- written from the prefetched values, no second round trip
                if (threadIdx.x == 0) {
                    ad.state[2 * (blockIdx.x + 1)] = b1p * ad.b1;
                    ad.state[2 * (blockIdx.x + 1) + 1] = b2p * ad.b2;
                }
            }
        }
        return;
    }
    // last block: scalars.  Everything thread 0 needs at the end is requested up front (one memory round trip, not three
    // dependent ones behind the reductions)
    double s0 = 0.0, s1 = 0.0, eps_th = 0.0, eps_m = 0.0, eps_v = 0.0;
    double old4[4] = {0.0, 0.0, 0.0, 0.0};     // pend: the previous iteration's reduced d-epsilon and losses, read before this one's overwrite them
    int hidx = -1, failed = 0, failed_now = 0;        // failed: the update this launch applies; failed_now: this iteration (the pad slot)
    if (threadIdx.x == 0 && xerr) { failed_now = *xerr; failed = pend ? xerr[1] : failed_now; }
    if (threadIdx.x == 0 && ad.theta) {
        s0 = ad.state[0]; s1 = ad.state[1];
        if (ad.hist) hidx = *ad.hist_idx;
        if (has_eps) { eps_th = ad.theta[P]; eps_m = ad.m[P]; eps_v = ad.v[P]; }
        if (pend) { old4[0] = has_eps ? RB[P] : 0.0; old4[1] = RB[Ptot]; old4[2] = RB[Ptot + 1]; old4[3] = RB[Ptot + 2]; }
    }
    double lv = 0.0, de = 0.0;
    for (long e = threadIdx.x; e < n_elem; e += blockDim.x) {
        lv += loss_e[e];
        if (has_eps && deps_e) de += deps_e[e];
    }
    double sq = 0.0;
    for (int i = threadIdx.x; i < n_data_part; i += blockDim.x) sq += data_part[i];
    lv = block_sum(lv, red);
    de = block_sum(de, red);
    sq = block_sum(sq, red);
    if (threadIdx.x == 0) {
        const double msq = n_data > 0 ? sq / (double)n_data : 0.0;
        const double eps_now = (has_eps && ad.theta) ? eps_th : 0.0;   // the coefficient this forward pass used
        if (has_eps && write_grad) {
            RB[P] = de;
            if (ad.theta && !failed) {   // the trainable epsilon (P3:63): adam_update with the operands fetched above
                const double ge = pend ? old4[0] : de;
                double mi, vi, ti;
                hpv_adam_one(ad.lr, ad.b1, ad.b2, ad.eps, s0, s1, ge, eps_m, eps_v, eps_th, mi, vi, ti);
                ad.m[P] = mi;
                ad.v[P] = vi;
                ad.theta[P] = ti;
            }
        }
        if (ad.theta && !failed) {
            ad.state[0] = s0 * ad.b1; ad.state[1] = s1 * ad.b2;
            if (ad.n_upd) *ad.n_upd += 1;
            if (ad.hist) {   // single-GPU training iteration: record this forward pass's loss (see AdamArgs)
                const int i = hidx;
                if (i >= 0 && i < ad.hist_cap) {   // the index saturates at hist_cap (it never wraps)
                    if (pend) { ad.hist[4 * i] = old4[1]; ad.hist[4 * i + 1] = old4[2]; ad.hist[4 * i + 2] = old4[3]; ad.hist[4 * i + 3] = eps_now; }
                    else
                    ad.hist[4 * i] = lv, ad.hist[4 * i + 1] = lossb_weight * msq, ad.hist[4 * i + 2] = msq, ad.hist[4 * i + 3] = eps_now;
                    *ad.hist_idx = i + 1;
                }
            }
        }
        RB[Ptot + 0] = lv;
        RB[Ptot + 1] = lossb_weight * msq;
        RB[Ptot + 2] = msq;
        RB[Ptot + 3] = failed_now ? 1.0 : 0.0;   // pad slot: the all-reduce carries a failure on any rank to every rank (k_adam)
        // launch counter of the shared-element kernels' tagged exchange (hpv_fused_dev.h): advanced HERE, behind the launch that used
        // the tag -- every workgroup of that launch has ended, so none of them can read the advanced value
        if (xiter_bump) *xiter_bump += 1u;
    }
}

void launch_finalize(const double* GPART_v, int rows_v, const double* GPART_b, int rows_b, const double* GPART_e,
                     int rows_e, const double* loss_e, long n_elem, const double* deps_e, const double* data_part,
                     int n_data_part, double lossb_weight, int n_data, int P, int has_eps, double* RB, int write_grad,
                     const AdamArgs* fused_adam, hipStream_t s, const int* xerr, unsigned int* xiter_bump, int pending_adam) {
    int gblocks = (P + FIN_COLS - 1) / FIN_COLS;
    AdamArgs ad{};
    if (fused_adam && write_grad) ad = *fused_adam;
    const int rows = (GPART_v ? rows_v : 0) + (GPART_b ? rows_b : 0) + (GPART_e ? rows_e : 0);
    if (rows <= 1024 && n_elem <= 4096)
        hipLaunchKernelGGL(k_finalize<256>, dim3(gblocks + 1), dim3(256), 0, s, GPART_v, rows_v, GPART_b, rows_b, GPART_e,
                           rows_e, loss_e, n_elem, deps_e, data_part, n_data_part, lossb_weight, n_data, P, has_eps, RB,
                           write_grad, ad, xerr, xiter_bump, pending_adam);
    else
        hipLaunchKernelGGL(k_finalize<1024>, dim3(gblocks + 1), dim3(1024), 0, s, GPART_v, rows_v, GPART_b, rows_b, GPART_e,
                           rows_e, loss_e, n_elem, deps_e, data_part, n_data_part, lossb_weight, n_data, P, has_eps, RB,
                           write_grad, ad, xerr, xiter_bump, pending_adam);
}
int adam_state_doubles(int P) { return 2 * ((P + FIN_COLS - 1) / FIN_COLS + 1); }

// ------------------------------------------------------------------------------------------------
// TF1 AdamOptimizer update (tf.train.AdamOptimizer(LR).minimize, P1:103-104):
//   lr_t = lr sqrt(1-b2^t)/(1-b1^t); m = b1 m + (1-b1) g; v = b2 v + (1-b2) g^2;
//   theta -= lr_t m / (sqrt(v) + eps)      -- eps OUTSIDE the bias correction, unlike torch.optim.Adam.
// state = {beta1^t, beta2^t} kept as running products exactly like TF's beta*_power variables.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(1024) k_adam(AdamArgs ad, const double* __restrict__ g, int P, int Ptot, int ncopies) {
    // The kernel is one latency chain behind the collective (the N > 1 iteration's tail): EVERY operand -- the failure flags, the
    // beta powers, this thread's parameter / moments / gradient, the history slot -- is requested up front, in ONE memory round
    // trip, instead of flag -> powers -> operands one after the other (round 5: the tail of the 1-rank-RCCL iteration, verdict item 1c).
    const int i0 = threadIdx.x;
    const bool own = i0 < Ptot;               // (Ptot <= 1024 on every instantiated network: one parameter per thread; more: the loop below)
    const double flag = g[Ptot + 3];
    const int xe = ad.xerr ? *ad.xerr : 0;
    const double b1p = ad.state[0], b2p = ad.state[1];
    double g0 = 0.0, m0 = 0.0, v0 = 0.0, t0 = 0.0;
    if (own) { g0 = g[i0]; m0 = ad.m[i0]; v0 = ad.v[i0]; t0 = ad.theta[i0]; }
    int hidx = -1;
    double h0 = 0.0, h1 = 0.0, h2 = 0.0, h3 = 0.0, sc0 = 0.0, sc1 = 0.0;
    if (threadIdx.x == 0 && ad.hist) { hidx = *ad.hist_idx; h0 = g[Ptot]; h1 = g[Ptot + 1]; h2 = g[Ptot + 2]; h3 = Ptot > P ? ad.theta[P] : 0.0; }
    const int c0 = (int)threadIdx.x - 512;    // the replicated copies of the beta powers: advanced by the upper half of the block
    if (c0 >= 0 && c0 < ncopies) { sc0 = ad.state[2 * c0]; sc1 = ad.state[2 * c0 + 1]; }
    // a failed SPLIT-mode barrier on ANY rank (pad slot of the all-reduced buffer: the sum of the ranks' flags; NaN counts) or
    // on this one: no update, no history entry, beta powers untouched -- and this rank's flag set, so that every rank reports it
    const bool failed = !(flag == 0.0) || xe;
    if (failed) {
        if (threadIdx.x == 0 && ad.xerr) *ad.xerr = 1;
        return;
    }
    // every up-front load above has returned in every wave before any wave stores below: thread 0's read of theta[P] (the history's
    // "epsilon before this update") must not race the store of the thread that owns index P in another wave (advisor, round 5;
    // `failed` is block-uniform, so the barrier is reached by all threads or by none)
    __syncthreads();
    if (threadIdx.x == 0 && ad.n_upd) *ad.n_upd += 1;
    if (threadIdx.x == 0 && ad.hist) {   // multi-GPU iteration: g is the all-reduced packed buffer, the losses follow the gradient
        if (hidx >= 0 && hidx < ad.hist_cap) {   // saturating index
            ad.hist[4 * hidx] = h0; ad.hist[4 * hidx + 1] = h1; ad.hist[4 * hidx + 2] = h2;
            ad.hist[4 * hidx + 3] = h3;          // epsilon before this update
            *ad.hist_idx = hidx + 1;
        }
    }
    if (own) {       // adam_update with the operands fetched above (same arithmetic, same order)
        double mi, vi, ti;
        hpv_adam_one(ad.lr, ad.b1, ad.b2, ad.eps, b1p, b2p, g0, m0, v0, t0, mi, vi, ti);
        ad.m[i0] = mi;
        ad.v[i0] = vi;
        ad.theta[i0] = ti;
    }
    for (int i = threadIdx.x + blockDim.x; i < Ptot; i += blockDim.x) adam_update(ad, i, g[i], b1p, b2p);
    // every replicated copy advances (each thread rewrites the copy it read above) -- behind a barrier: every wave has read copy 0
    // (its b1p / b2p) by then
    __syncthreads();
    if (c0 >= 0) {
        if (c0 < ncopies) { ad.state[2 * c0] = sc0 * ad.b1; ad.state[2 * c0 + 1] = sc1 * ad.b2; }
        for (int c = c0 + 512; c < ncopies; c += 512) { ad.state[2 * c] *= ad.b1; ad.state[2 * c + 1] *= ad.b2; }   // (> 512 copies: wide networks)
    }
}

// ------------------------------------------------------------------------------------------------
// One-shot all-reduce(sum) of the packed buffer over peer-mapped mailboxes, fused with the TF1 Adam update
// (multi-GPU iteration = forward, reverse, finalize, THIS; no collective library call inside the iteration).
//   exchange k, parity p = k & 1:   write RB into inbox[r][p][rank] of every rank r (own included),
//   system-scope fence, release-store k+1 into flag[r][p][rank]; wait until own flag[p][r] >= k+1 for every r;
//   RB = sum over r of inbox[own][p][r] in rank order (identical bits on every rank).
// Two parities make the mailbox safe: a rank can only start exchange k+2 after every peer has contributed to k+1,
// i.e. after every peer has finished reading exchange k.  The wait is bounded (err flag instead of a hang).
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(1024) k_p2p_exchange(P2PArgs pp, double* __restrict__ RB, AdamArgs ad, int P, int Ptot,
                                                      int ncopies) {
    __shared__ int s_fail;
    if (threadIdx.x == 0) s_fail = *pp.err;      // sticky: once an exchange has failed, every later one is a no-op
    __syncthreads();
    if (s_fail) return;
    const unsigned long long k = *pp.counter;
    const int par = (int)(k & 1ULL), W = pp.world, n = pp.n;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const double v = RB[i];
        for (int r = 0; r < W; ++r)
            __hip_atomic_store(pp.inbox[r] + ((long)par * W + pp.rank) * n + i, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    __threadfence_system();
    __syncthreads();
    if ((int)threadIdx.x < W)
        __hip_atomic_store(pp.flag[threadIdx.x] + par * W + pp.rank, k + 1ULL, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    if ((int)threadIdx.x < W) {
        // Bounded by WALL CLOCK (s_memrealtime counts at 100 MHz), not by a spin count.  A peer that gave up publishes
        // HPV_P2P_POISON instead of its arrival count, so that every rank aborts the same exchange.
        const unsigned long long* f = pp.flag[pp.rank] + par * W + threadIdx.x;
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        for (;;) {
            const unsigned long long v = __hip_atomic_load(f, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM);
            if (v == HPV_P2P_POISON) { atomicExch(&s_fail, 1); break; }
            if (v >= k + 1ULL) break;
            if (__builtin_amdgcn_s_memrealtime() - t0 > pp.timeout_ticks) { atomicExch(&s_fail, 1); break; }
            __builtin_amdgcn_s_sleep(2);
        }
    }
    __syncthreads();
    if (s_fail) {
        // leave RB, theta, m, v, the beta powers and the exchange counter untouched; tell the host and the peers
        if (threadIdx.x == 0) *pp.err = 1;
        if ((int)threadIdx.x < W) {
            __hip_atomic_store(pp.flag[threadIdx.x] + 0 * W + pp.rank, HPV_P2P_POISON, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(pp.flag[threadIdx.x] + 1 * W + pp.rank, HPV_P2P_POISON, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        return;
    }
    const double* mine = pp.inbox[pp.rank] + (long)par * W * n;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        double sum = 0.0;
        for (int r = 0; r < W; ++r) sum += __hip_atomic_load(mine + (long)r * n + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        RB[i] = sum;
    }
    __syncthreads();
    if (threadIdx.x == 0) *pp.counter = k + 1ULL;
    if (!ad.theta) return;
    if (!(RB[Ptot + 3] == 0.0) || (ad.xerr && *ad.xerr)) {   // a SPLIT-mode barrier failed on some rank: see k_adam
        if (threadIdx.x == 0 && ad.xerr) *ad.xerr = 1;
        return;
    }
    // ---- TF1 Adam on the reduced gradient (same as k_adam) ----
    const double b1p = ad.state[0], b2p = ad.state[1];
    if (threadIdx.x == 0 && ad.n_upd) *ad.n_upd += 1;
    if (threadIdx.x == 0 && ad.hist) {
        const int i = *ad.hist_idx;
        if (i >= 0 && i < ad.hist_cap) {   // saturating index
            ad.hist[4 * i] = RB[Ptot]; ad.hist[4 * i + 1] = RB[Ptot + 1]; ad.hist[4 * i + 2] = RB[Ptot + 2];
            ad.hist[4 * i + 3] = Ptot > P ? ad.theta[P] : 0.0;
            *ad.hist_idx = i + 1;
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < Ptot; i += blockDim.x) adam_update(ad, i, RB[i], b1p, b2p);
    __syncthreads();
    for (int c = threadIdx.x; c < ncopies; c += blockDim.x) {
        ad.state[2 * c] *= ad.b1;
        ad.state[2 * c + 1] *= ad.b2;
    }
}
void launch_p2p_exchange(const P2PArgs& pp, double* RB, const AdamArgs* adam_or_null, int P, int Ptot, hipStream_t s) {
    AdamArgs ad{};
    if (adam_or_null) ad = *adam_or_null;
    hipLaunchKernelGGL(k_p2p_exchange, dim3(1), dim3(1024), 0, s, pp, RB, ad, P, Ptot, adam_state_doubles(P) / 2);
}

void launch_adam(const AdamArgs& ad, const double* RB, int P, int Ptot, hipStream_t s) {
    hipLaunchKernelGGL(k_adam, dim3(1), dim3(1024), 0, s, ad, RB, P, Ptot, adam_state_doubles(P) / 2);
}

// ------------------------------------------------------------------------------------------------
// Test hook: the device activation (value, derivative) and ocml's own, element-wise.
// ------------------------------------------------------------------------------------------------
__global__ void k_debug_act(int act, const double* __restrict__ x, int n, double* __restrict__ a,
                            double* __restrict__ a1, double* __restrict__ ref) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double v, d;
    act_eval(act, x[i], v, d);
    a[i] = v;
    a1[i] = d;
    ref[i] = (act == HPV_ACT_TANH) ? tanh(x[i]) : sin(x[i]);
}
void launch_debug_act(int act, const double* x, int n, double* a, double* a1, double* ref, hipStream_t s) {
    hipLaunchKernelGGL(k_debug_act, dim3((n + 255) / 256), dim3(256), 0, s, act, x, n, a, a1, ref);
}

// ------------------------------------------------------------------------------------------------
// Strong-form PINN residual of the 2-D Poisson problem (P2:187-194): r = u_xx + u_yy - f at the
// collocation points; lossp = mean(r^2) (P2:124).  Channels: [u, u_x, u_y, u_xx + u_yy] (the Laplacian as one mixed second tangent).
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_pinn_residual(const double* __restrict__ OUT, const double* __restrict__ f,
                                                      double* __restrict__ GBAR, double* __restrict__ part, long N,
                                                      int n, long n_total, int write_gbar) {
    // n points of THIS shard; the mean of P2:124 runs over the n_total collocation points of all shards
    __shared__ double red[16];
    double sq = 0.0;
    const double sc = 2.0 / (double)n_total;
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < n; p += gridDim.x * blockDim.x) {
        const double r = OUT[3 * N + p] - f[p];
        sq += r * r;
        if (write_gbar) GBAR[3 * N + p] = sc * r;
    }
    sq = block_sum(sq, red);
    if (threadIdx.x == 0) part[blockIdx.x] = sq / (double)n_total;
}
int pinn_residual_parts(int n) { int b = (n + 255) / 256; return b > 64 ? 64 : (b < 1 ? 1 : b); }
void launch_pinn_residual(const double* OUT, const double* f, double* GBAR, double* part, long N, int n, long n_total,
                          int write_gbar, hipStream_t s) {
    hipLaunchKernelGGL(k_pinn_residual, dim3(pinn_residual_parts(n)), dim3(256), 0, s, OUT, f, GBAR, part, N, n, n_total, write_gbar);
}

// ------------------------------------------------------------------------------------------------
// Driver-side table generation on the device: Jacobi polynomials by the three-term
// recurrence (never expanded coefficients: those are off by 1e13 at n = 61), the Gauss-Lobatto-Legendre rule
// (GaussLobattoJacobiWeights(Q, 0, 0), Q:47-61) by Newton from the Chebyshev-Gauss-Lobatto points, and the
// test-function tables phi_n = P_{n+1} - P_{n-1} with their first two derivatives (Test_fcn / dTest_fcn,
// P1:157-183).
// ------------------------------------------------------------------------------------------------
__device__ double dev_jacobi(int n, double a, double b, double x) {
    if (n < 0) return 0.0;
    double p0 = 1.0;
    if (n == 0) return p0;
    double p1 = 0.5 * ((a - b) + (a + b + 2.0) * x);
    for (int k = 1; k < n; ++k) {
        const double c = 2.0 * k + a + b;
        const double a1 = 2.0 * (k + 1.0) * (k + a + b + 1.0) * c;
        const double a2 = (c + 1.0) * (a * a - b * b);
        const double a3 = c * (c + 1.0) * (c + 2.0);
        const double a4 = 2.0 * (k + a) * (k + b) * (c + 2.0);
        const double p2 = ((a2 + a3 * x) * p1 - a4 * p0) / a1;
        p0 = p1;
        p1 = p2;
    }
    return p1;
}

// interior GLL nodes = zeros of P'_{Q-1} = zeros of P_{Q-2}^{(1,1)}; weights 2 / (Q (Q-1) P_{Q-1}(x)^2)
__global__ void k_gll_rule(int Q, double* __restrict__ x, double* __restrict__ w) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= Q) return;
    double xk;
    if (k == 0) xk = -1.0;
    else if (k == Q - 1) xk = 1.0;
    else {
        xk = -cos(3.14159265358979323846 * (double)k / (double)(Q - 1));
        for (int it = 0; it < 50; ++it) {
            const double p = dev_jacobi(Q - 2, 1.0, 1.0, xk);
            const double dp = 0.5 * (Q + 1.0) * dev_jacobi(Q - 3, 2.0, 2.0, xk);   // d/dx P_n^{(1,1)} = (n+3)/2 P_{n-1}^{(2,2)}
            const double dx = p / dp;
            xk -= dx;
            if (fabs(dx) < 1e-16) break;
        }
    }
    const double pl = dev_jacobi(Q - 1, 0.0, 0.0, xk);
    x[k] = xk;
    w[k] = 2.0 / ((double)Q * (Q - 1.0) * pl * pl);
}

__global__ void k_test_tables(int ntest, int q, const double* __restrict__ xi, double* __restrict__ tab) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= ntest * q) return;
    const int n = idx / q + 1;
    const double x = xi[idx % q];
    const double t0 = dev_jacobi(n + 1, 0.0, 0.0, x) - dev_jacobi(n - 1, 0.0, 0.0, x);
    double t1 = 0.5 * (n + 2.0) * dev_jacobi(n, 1.0, 1.0, x);
    if (n >= 2) t1 -= 0.5 * n * dev_jacobi(n - 2, 1.0, 1.0, x);
    double t2 = 0.25 * (n + 2.0) * (n + 3.0) * dev_jacobi(n - 1, 2.0, 2.0, x);
    if (n >= 3) t2 -= 0.25 * n * (n + 1.0) * dev_jacobi(n - 3, 2.0, 2.0, x);
    tab[idx] = t0;
    tab[(long)ntest * q + idx] = t1;
    tab[2L * ntest * q + idx] = t2;
}
void launch_gll_rule(int Q, double* x, double* w, hipStream_t s) {
    hipLaunchKernelGGL(k_gll_rule, dim3((Q + 63) / 64), dim3(64), 0, s, Q, x, w);
}
void launch_test_tables(int ntest, int q, const double* xi, double* tab, hipStream_t s) {
    hipLaunchKernelGGL(k_test_tables, dim3((ntest * q + 255) / 256), dim3(256), 0, s, ntest, q, xi, tab);
}
