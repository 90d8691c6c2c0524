/* cpu_closed_form.c -- TEST INFRASTRUCTURE / CPU BASELINE, NOT PRODUCT CODE.
 *
 * Only tests/, __graft_entry__.smoke()/build() and bench.py's cpu_baseline leg may build, load or call this file
 * (same rule as oracle/vpinn_oracle.py).  The product (hp_vpinns_amd/) never links or loads it.
 *
 * What it is: baseline "B" of BASELINE.md section 3 -- a plain C / OpenMP restatement of one training iteration of
 * the reference's Poisson-2D hp-VPINN with var_form 1 (the BASELINE config-3 / config-4 loss graph):
 *
 *   network          neural_net / net_u / net_dxu / net_dyu        main/Poisson-2D/hp-VPINN-Poisson-2D.py:158-185
 *   element maps     x = g_e + (g_{e+1}-g_e)/2 (xi+1), jacobians    P2:75-79
 *   residual         U[k][r] = -(J/Jx) sum wx phi'_r wy phi_k u_x - (J/Jy) sum wx phi_r wy phi'_k u_y   P2:98-105
 *                    R = U - F_ext[ex,ey],  loss_e = mean(R^2), summed over elements                     P2:117-120
 *   boundary term    lossb = mean((u_d - u_NN)^2), loss = 10 lossb + lossv                               P2:122-127
 *   optimiser        tf.train.AdamOptimizer(LR).minimize (TF1 rule: epsilon OUTSIDE the bias correction) P2:131-132
 *
 * in closed form: Taylor-mode channels (u, u_x, u_y) pushed forward through the MLP, sum-factorised projection, the
 * hand-derived reverse pass.  Elements are distributed over OpenMP threads; each thread keeps its own gradient row,
 * rows are added in thread order (deterministic for a fixed thread count).  It is checked against the autograd
 * oracle in tests/test_oracle.py (loss <= 1e-12, gradient <= 1e-10 relative) before anything is timed with it.
 *
 * Build (oracle/build_cpu_baseline.sh):  gcc -O3 -march=x86-64-v3 -fopenmp -shared -fPIC cpu_closed_form.c -lm
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define MAXL 8     /* hidden layers */
#define MAXH 64    /* hidden width  */

typedef struct {
    int d, nh, H;                 /* inputs (2), hidden layers, width */
    const double *W[MAXL + 1], *b[MAXL + 1];
    int woff[MAXL + 1], boff[MAXL + 1];
    int P;
} Net;

static int net_init(Net* n, const double* theta, const int* layers, int n_layers) {
    if (n_layers < 3 || n_layers - 2 > MAXL || layers[0] != 2 || layers[n_layers - 1] != 1) return -1;
    n->d = 2; n->nh = n_layers - 2; n->H = layers[1];
    if (n->H > MAXH) return -1;
    int off = 0;
    for (int l = 0; l < n_layers - 1; ++l) {
        if (l > 0 && l < n_layers - 1 && layers[l] != n->H) return -1;
        n->woff[l] = off; n->W[l] = theta + off; off += layers[l] * layers[l + 1];
        n->boff[l] = off; n->b[l] = theta + off; off += layers[l + 1];
    }
    n->P = off;
    return 0;
}

/* workspace of one batch of n points: per hidden layer s, zx, zy as [H][n] (point index fastest) */
typedef struct {
    int n, cap;
    double *s, *zx, *zy;          /* [nh][H][cap] */
    double *u, *ux, *uy;          /* [cap] */
    double *t0, *t1, *t2, *t3, *t4, *t5;   /* [H][cap] scratch */
} Work;

static int work_alloc(Work* w, int cap, int nh, int H) {
    size_t a = (size_t)nh * H * cap;
    w->cap = cap;
    w->s = malloc(3 * a * sizeof(double));
    w->u = malloc(3 * (size_t)cap * sizeof(double));
    w->t0 = malloc(6 * (size_t)H * cap * sizeof(double));
    if (!w->s || !w->u || !w->t0) return -1;
    w->zx = w->s + a; w->zy = w->zx + a;
    w->ux = w->u + cap; w->uy = w->ux + cap;
    w->t1 = w->t0 + (size_t)H * cap; w->t2 = w->t1 + (size_t)H * cap; w->t3 = w->t2 + (size_t)H * cap;
    w->t4 = w->t3 + (size_t)H * cap; w->t5 = w->t4 + (size_t)H * cap;
    return 0;
}
static void work_free(Work* w) { free(w->s); free(w->u); free(w->t0); }

/* forward: channels (u, u_x, u_y) at the n points (x[p], y[p]) */
static void forward(const Net* N, Work* w, const double* x, const double* y, int n) {
    const int H = N->H, cap = w->cap;
    w->n = n;
    for (int l = 0; l < N->nh; ++l) {
        double* s = w->s + (size_t)l * H * cap;
        double* zx = w->zx + (size_t)l * H * cap;
        double* zy = w->zy + (size_t)l * H * cap;
        if (l == 0) {
            for (int j = 0; j < H; ++j) {
                const double w0 = N->W[0][j], w1 = N->W[0][H + j], bj = N->b[0][j];
                double* sj = s + (size_t)j * cap; double* zxj = zx + (size_t)j * cap; double* zyj = zy + (size_t)j * cap;
                for (int p = 0; p < n; ++p) { sj[p] = tanh(bj + x[p] * w0 + y[p] * w1); zxj[p] = w0; zyj[p] = w1; }
            }
        } else {
            const double* sp = w->s + (size_t)(l - 1) * H * cap;
            const double* zxp = w->zx + (size_t)(l - 1) * H * cap;
            const double* zyp = w->zy + (size_t)(l - 1) * H * cap;
            /* inputs of this layer: h = s_prev, h_x = (1 - s_prev^2) zx_prev, h_y likewise */
            double *hx = w->t0, *hy = w->t1;
            for (int i = 0; i < H; ++i)
                for (int p = 0; p < n; ++p) {
                    const double sv = sp[(size_t)i * cap + p], d1 = 1.0 - sv * sv;
                    hx[(size_t)i * cap + p] = d1 * zxp[(size_t)i * cap + p];
                    hy[(size_t)i * cap + p] = d1 * zyp[(size_t)i * cap + p];
                }
            for (int j = 0; j < H; ++j) {
                double* sj = s + (size_t)j * cap; double* zxj = zx + (size_t)j * cap; double* zyj = zy + (size_t)j * cap;
                const double bj = N->b[l][j];
                for (int p = 0; p < n; ++p) { sj[p] = bj; zxj[p] = 0.0; zyj[p] = 0.0; }
                for (int i = 0; i < H; ++i) {
                    const double wij = N->W[l][i * H + j];
                    const double* hi = sp + (size_t)i * cap; const double* hxi = hx + (size_t)i * cap; const double* hyi = hy + (size_t)i * cap;
                    for (int p = 0; p < n; ++p) { sj[p] += wij * hi[p]; zxj[p] += wij * hxi[p]; zyj[p] += wij * hyi[p]; }
                }
                for (int p = 0; p < n; ++p) sj[p] = tanh(sj[p]);
            }
        }
    }
    {   /* linear head */
        const int l = N->nh;
        const double* sp = w->s + (size_t)(l - 1) * H * cap;
        const double* zxp = w->zx + (size_t)(l - 1) * H * cap;
        const double* zyp = w->zy + (size_t)(l - 1) * H * cap;
        const double bo = N->b[l][0];
        for (int p = 0; p < n; ++p) { w->u[p] = bo; w->ux[p] = 0.0; w->uy[p] = 0.0; }
        for (int i = 0; i < H; ++i) {
            const double wi = N->W[l][i];
            for (int p = 0; p < n; ++p) {
                const double sv = sp[(size_t)i * cap + p], d1 = 1.0 - sv * sv;
                w->u[p] += wi * sv; w->ux[p] += wi * d1 * zxp[(size_t)i * cap + p]; w->uy[p] += wi * d1 * zyp[(size_t)i * cap + p];
            }
        }
    }
}

/* reverse: g += d(sum_p gu u + gx u_x + gy u_y)/d theta, from the workspace of the last forward() */
static void backward(const Net* N, Work* w, const double* x, const double* y, const double* gu, const double* gx,
                     const double* gy, double* g) {
    const int H = N->H, cap = w->cap, n = w->n, L = N->nh;
    double *hb = w->t0, *hxb = w->t1, *hyb = w->t2, *zb = w->t3, *zxb = w->t4, *zyb = w->t5;
    {   /* head */
        const double* sp = w->s + (size_t)(L - 1) * H * cap;
        const double* zxp = w->zx + (size_t)(L - 1) * H * cap;
        const double* zyp = w->zy + (size_t)(L - 1) * H * cap;
        double dbo = 0.0;
        for (int p = 0; p < n; ++p) dbo += gu[p];
        g[N->boff[L]] += dbo;
        for (int i = 0; i < H; ++i) {
            const double wi = N->W[L][i];
            double acc = 0.0;
            for (int p = 0; p < n; ++p) {
                const double sv = sp[(size_t)i * cap + p], d1 = 1.0 - sv * sv;
                acc += sv * gu[p] + d1 * zxp[(size_t)i * cap + p] * gx[p] + d1 * zyp[(size_t)i * cap + p] * gy[p];
                hb[(size_t)i * cap + p] = gu[p] * wi; hxb[(size_t)i * cap + p] = gx[p] * wi; hyb[(size_t)i * cap + p] = gy[p] * wi;
            }
            g[N->woff[L] + i] += acc;
        }
    }
    for (int l = L - 1; l >= 0; --l) {
        const double* s = w->s + (size_t)l * H * cap;
        const double* zx = w->zx + (size_t)l * H * cap;
        const double* zy = w->zy + (size_t)l * H * cap;
        for (int j = 0; j < H; ++j) {
            double db = 0.0;
            for (int p = 0; p < n; ++p) {
                const size_t k = (size_t)j * cap + p;
                const double sv = s[k], d1 = 1.0 - sv * sv, d2 = -2.0 * sv * d1;
                zxb[k] = hxb[k] * d1; zyb[k] = hyb[k] * d1;
                zb[k] = hb[k] * d1 + hxb[k] * d2 * zx[k] + hyb[k] * d2 * zy[k];
                db += zb[k];
            }
            g[N->boff[l] + j] += db;
        }
        if (l == 0) {
            for (int j = 0; j < H; ++j) {
                double a0 = 0.0, a1 = 0.0;
                for (int p = 0; p < n; ++p) {
                    const size_t k = (size_t)j * cap + p;
                    a0 += x[p] * zb[k] + zxb[k]; a1 += y[p] * zb[k] + zyb[k];
                }
                g[N->woff[0] + j] += a0; g[N->woff[0] + H + j] += a1;
            }
        } else {
            const double* sp = s - (size_t)H * cap;      /* layer l-1 */
            const double* zxp = zx - (size_t)H * cap;
            const double* zyp = zy - (size_t)H * cap;
            /* dW[i][j] and the adjoint of this layer's inputs (written over hb/hxb/hyb after they were consumed) */
            for (int i = 0; i < H; ++i) {
                double acc[MAXH];
                for (int j = 0; j < H; ++j) acc[j] = 0.0;
                for (int j = 0; j < H; ++j) {
                    double a = 0.0;
                    for (int p = 0; p < n; ++p) {
                        const size_t ki = (size_t)i * cap + p, kj = (size_t)j * cap + p;
                        const double sv = sp[ki], d1 = 1.0 - sv * sv;
                        a += sv * zb[kj] + d1 * zxp[ki] * zxb[kj] + d1 * zyp[ki] * zyb[kj];
                    }
                    acc[j] = a;
                }
                for (int j = 0; j < H; ++j) g[N->woff[l] + i * H + j] += acc[j];
            }
            for (int i = 0; i < H; ++i) {
                double* hbi = hb + (size_t)i * cap; double* hxbi = hxb + (size_t)i * cap; double* hybi = hyb + (size_t)i * cap;
                for (int p = 0; p < n; ++p) { hbi[p] = 0.0; hxbi[p] = 0.0; hybi[p] = 0.0; }
                for (int j = 0; j < H; ++j) {
                    const double wij = N->W[l][i * H + j];
                    const double* zbj = zb + (size_t)j * cap; const double* zxbj = zxb + (size_t)j * cap; const double* zybj = zyb + (size_t)j * cap;
                    for (int p = 0; p < n; ++p) { hbi[p] += wij * zbj[p]; hxbi[p] += wij * zxbj[p]; hybi[p] += wij * zybj[p]; }
                }
            }
        }
    }
}

typedef struct {
    int Q, ntx, nty, nex, ney, nd;
    const double *xi, *wq, *gridx, *gridy, *tabx, *taby, *F, *Xd, *ud;   /* tab*: [2][nt][Q] = phi, phi' */
    double lossb_weight;
} Problem;

/* per-thread scratch, allocated once per call of hpvc_loss_grad / hpvc_train (not per iteration) */
typedef struct {
    int nt, P, cap, ok;
    Work* w;                     /* [nt] */
    double *x, *gb, *T;          /* [nt][2 cap], [nt][3 cap], [nt][Q ntx + NR + nty Q] */
    double *grows, *lrows;       /* [nt][P], [nt] */
    size_t tsz;
} Scratch;

static int resolve_threads(int nthreads) {
#ifdef _OPENMP
    return nthreads > 0 ? nthreads : omp_get_max_threads();
#else
    (void)nthreads;
    return 1;
#endif
}

static void scratch_free(Scratch* sc) {
    if (sc->w) { for (int t = 0; t < sc->nt; ++t) if (sc->w[t].s) work_free(&sc->w[t]); free(sc->w); }
    free(sc->x); free(sc->gb); free(sc->T); free(sc->grows); free(sc->lrows);
}

static int scratch_alloc(Scratch* sc, const Problem* pb, const Net* N, int nt) {
    const int NQ = pb->Q * pb->Q, NR = pb->ntx * pb->nty;
    memset(sc, 0, sizeof *sc);
    sc->nt = nt; sc->P = N->P; sc->cap = NQ > pb->nd ? NQ : pb->nd;
    sc->tsz = (size_t)pb->Q * pb->ntx + NR + (size_t)pb->nty * pb->Q;
    sc->w = calloc((size_t)nt, sizeof(Work));
    sc->x = malloc((size_t)nt * 2 * sc->cap * sizeof(double));
    sc->gb = malloc((size_t)nt * 3 * sc->cap * sizeof(double));
    sc->T = malloc((size_t)nt * sc->tsz * sizeof(double));
    sc->grows = malloc((size_t)nt * N->P * sizeof(double));
    sc->lrows = malloc((size_t)nt * sizeof(double));
    int bad = !sc->w || !sc->x || !sc->gb || !sc->T || !sc->grows || !sc->lrows;
    for (int t = 0; t < nt && !bad; ++t) bad = work_alloc(&sc->w[t], sc->cap, N->nh, N->H);
    if (bad) { scratch_free(sc); return -2; }
    return 0;
}

/* loss (3) and gradient (P) of the whole problem at theta */
static int loss_grad(const Problem* pb, const double* theta, const int* layers, int n_layers, Scratch* sc,
                     double* loss3, double* grad) {
    Net N;
    if (net_init(&N, theta, layers, n_layers)) return -1;
    const int Q = pb->Q, NQ = Q * Q, ntx = pb->ntx, nty = pb->nty, NR = ntx * nty;
    const int ne = pb->nex * pb->ney, P = N.P, nt = sc->nt;
    double* grows = sc->grows;
    double* lrows = sc->lrows;
    double msq = 0.0;
#pragma omp parallel num_threads(nt)
    {
        int tid = 0;
#ifdef _OPENMP
        tid = omp_get_thread_num();
#endif
        const int cap = sc->cap;
        Work w = sc->w[tid];
        double* x = sc->x + (size_t)tid * 2 * cap;
        double* gb = sc->gb + (size_t)tid * 3 * cap;
        double* T = sc->T + (size_t)tid * sc->tsz;
        {
            double* y = x + cap;
            double *gu = gb, *gx = gb + cap, *gy = gb + 2 * cap;
            double* U = T + (size_t)Q * ntx;
            double* S = U + NR;
            double* g = grows + (size_t)tid * P;
            memset(g, 0, (size_t)P * sizeof(double));
            lrows[tid] = 0.0;
#pragma omp for schedule(static)
            for (int e = 0; e < ne; ++e) {
                const int ex = e / pb->ney, ey = e % pb->ney;
                const double gx0 = pb->gridx[ex], gx1 = pb->gridx[ex + 1], gy0 = pb->gridy[ey], gy1 = pb->gridy[ey + 1];
                for (int j = 0; j < Q; ++j)
                    for (int i = 0; i < Q; ++i) {                     /* q = j*Q + i, x fastest (P2:362-365) */
                        x[j * Q + i] = gx0 + (gx1 - gx0) / 2 * (pb->xi[i] + 1);
                        y[j * Q + i] = gy0 + (gy1 - gy0) / 2 * (pb->xi[j] + 1);
                    }
                const double Jx = (gx1 - gx0) / 2, Jy = (gy1 - gy0) / 2, J = Jx * Jy;
                const double c[2] = {-(J / Jx), -(J / Jy)};
                forward(&N, &w, x, y, NQ);
                for (int o = 0; o < NR; ++o) U[o] = -pb->F[(size_t)e * NR + o];
                for (int t = 0; t < 2; ++t) {
                    const double* G = t == 0 ? w.ux : w.uy;
                    const double* A = pb->tabx + (size_t)(t == 0 ? 1 : 0) * ntx * Q;   /* phi'_r (t=0) / phi_r */
                    const double* B = pb->taby + (size_t)(t == 0 ? 0 : 1) * nty * Q;   /* phi_k / phi'_k (t=1) */
                    for (int j = 0; j < Q; ++j)
                        for (int r = 0; r < ntx; ++r) {
                            double a = 0.0;
                            for (int i = 0; i < Q; ++i) a += pb->wq[i] * A[r * Q + i] * G[j * Q + i];
                            T[j * ntx + r] = a;
                        }
                    for (int k = 0; k < nty; ++k)
                        for (int r = 0; r < ntx; ++r) {
                            double a = 0.0;
                            for (int j = 0; j < Q; ++j) a += pb->wq[j] * B[k * Q + j] * T[j * ntx + r];
                            U[k * ntx + r] += c[t] * a;
                        }
                }
                double sq = 0.0;
                for (int o = 0; o < NR; ++o) sq += U[o] * U[o];
                lrows[tid] += sq / NR;
                if (grad) {
                    for (int p = 0; p < NQ; ++p) gu[p] = 0.0;
                    for (int t = 0; t < 2; ++t) {
                        double* Gb = t == 0 ? gx : gy;
                        const double* A = pb->tabx + (size_t)(t == 0 ? 1 : 0) * ntx * Q;
                        const double* B = pb->taby + (size_t)(t == 0 ? 0 : 1) * nty * Q;
                        for (int k = 0; k < nty; ++k)
                            for (int i = 0; i < Q; ++i) {
                                double a = 0.0;
                                for (int r = 0; r < ntx; ++r) a += pb->wq[i] * A[r * Q + i] * U[k * ntx + r];
                                S[k * Q + i] = a * (2.0 / NR) * c[t];
                            }
                        for (int j = 0; j < Q; ++j)
                            for (int i = 0; i < Q; ++i) {
                                double a = 0.0;
                                for (int k = 0; k < nty; ++k) a += pb->wq[j] * B[k * Q + j] * S[k * Q + i];
                                Gb[j * Q + i] = a;
                            }
                    }
                    backward(&N, &w, x, y, gu, gx, gy, g);
                }
            }
#pragma omp single
            {
                /* boundary term on one thread (P2:122): a few hundred points */
                const int nd = pb->nd;
                if (nd > 0) {
                    for (int p = 0; p < nd; ++p) { x[p] = pb->Xd[2 * p]; y[p] = pb->Xd[2 * p + 1]; }
                    forward(&N, &w, x, y, nd);
                    double sq = 0.0;
                    for (int p = 0; p < nd; ++p) {
                        const double dd = pb->ud[p] - w.u[p];
                        sq += dd * dd;
                        gu[p] = -2.0 * pb->lossb_weight / nd * dd; gx[p] = 0.0; gy[p] = 0.0;
                    }
                    msq = sq / nd;
                    if (grad) backward(&N, &w, x, y, gu, gx, gy, g);
                }
            }
        }
    }
    double lossv = 0.0;
    for (int t = 0; t < nt; ++t) lossv += lrows[t];
    if (grad) {
        for (int i = 0; i < P; ++i) {
            double a = 0.0;
            for (int t = 0; t < nt; ++t) a += grows[(size_t)t * P + i];
            grad[i] = a;
        }
    }
    loss3[0] = pb->lossb_weight * msq + lossv; loss3[1] = msq; loss3[2] = lossv;
    return 0;
}

static void fill(Problem* pb, const double* xi, const double* wq, int Q, const double* gridx, int nex,
                 const double* gridy, int ney, const double* tabx, int ntx, const double* taby, int nty,
                 const double* F, const double* Xd, const double* ud, int nd, double lossb_weight) {
    pb->Q = Q; pb->ntx = ntx; pb->nty = nty; pb->nex = nex; pb->ney = ney; pb->nd = nd;
    pb->xi = xi; pb->wq = wq; pb->gridx = gridx; pb->gridy = gridy; pb->tabx = tabx; pb->taby = taby;
    pb->F = F; pb->Xd = Xd; pb->ud = ud; pb->lossb_weight = lossb_weight;
}

int hpvc_loss_grad(const double* theta, const int* layers, int n_layers, const double* xi, const double* wq, int Q,
                   const double* gridx, int nex, const double* gridy, int ney, const double* tabx, int ntx,
                   const double* taby, int nty, const double* F, const double* Xd, const double* ud, int nd,
                   double lossb_weight, int nthreads, double* loss3, double* grad) {
    Problem pb;
    fill(&pb, xi, wq, Q, gridx, nex, gridy, ney, tabx, ntx, taby, nty, F, Xd, ud, nd, lossb_weight);
    Net N;
    if (net_init(&N, theta, layers, n_layers)) return -1;
    Scratch sc;
    if (scratch_alloc(&sc, &pb, &N, resolve_threads(nthreads))) return -2;
    const int rc = loss_grad(&pb, theta, layers, n_layers, &sc, loss3, grad);
    scratch_free(&sc);
    return rc;
}

/* n_iters TF1-Adam iterations in place (theta, m, v, bpow = {beta1^t, beta2^t}); loss_hist[3*it] = the loss triple
 * of the forward pass of iteration it (i.e. BEFORE its update), or NULL */
int hpvc_train(double* theta, double* m, double* v, double* bpow, const int* layers, int n_layers, const double* xi,
               const double* wq, int Q, const double* gridx, int nex, const double* gridy, int ney, const double* tabx,
               int ntx, const double* taby, int nty, const double* F, const double* Xd, const double* ud, int nd,
               double lossb_weight, double lr, int nthreads, int n_iters, double* loss_hist) {
    Problem pb;
    fill(&pb, xi, wq, Q, gridx, nex, gridy, ney, tabx, ntx, taby, nty, F, Xd, ud, nd, lossb_weight);
    Net N;
    if (net_init(&N, theta, layers, n_layers)) return -1;
    Scratch sc;
    if (scratch_alloc(&sc, &pb, &N, resolve_threads(nthreads))) return -2;
    double* g = malloc((size_t)N.P * sizeof(double));
    if (!g) { scratch_free(&sc); return -2; }
    const double b1 = 0.9, b2 = 0.999, eps = 1e-8;
    int rc = 0;
    for (int it = 0; it < n_iters && !rc; ++it) {
        double l3[3];
        rc = loss_grad(&pb, theta, layers, n_layers, &sc, l3, g);
        if (rc) break;
        if (loss_hist) { loss_hist[3 * it] = l3[0]; loss_hist[3 * it + 1] = l3[1]; loss_hist[3 * it + 2] = l3[2]; }
        const double lr_t = lr * sqrt(1.0 - bpow[1]) / (1.0 - bpow[0]);
        for (int i = 0; i < N.P; ++i) {
            m[i] = b1 * m[i] + (1.0 - b1) * g[i];
            v[i] = b2 * v[i] + (1.0 - b2) * g[i] * g[i];
            theta[i] -= lr_t * m[i] / (sqrt(v[i]) + eps);
        }
        bpow[0] *= b1; bpow[1] *= b2;
    }
    free(g);
    scratch_free(&sc);
    return rc;
}

int hpvc_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
