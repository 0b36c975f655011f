// C ABI of the phase-diversity estimator (include/fastmpc.h: fmpc_est_*): handle with the constant operands in HBM, one
// call per batch of residual phase screens.  Kernels: fmpc_kernel_estimator.hip; host builders: fmpc_host.cpp.
#include <hip/hip_runtime.h>
#include <math.h>
#include <mutex>
#include <new>
#include <vector>
#include "../../include/fastmpc.h"
#include "fmpc_estimator.h"
#include "fmpc_host.h"
#include "fmpc_alloc.h"                    // counted hipMalloc / hipFree: the estimator's buffers move fmpc_alloc_generation too

struct fmpc_est_s {
    int device, len, d, first, ndiv, nx, p, rank;
    double scale;
    double* pool;                        // D_re | D_im | Fimg | G | b_s
    size_t oDre, oDim, oF, oG, ob;
    int* qlist;                          // [len / 16][2]: range of the k-steps inside the pupil per row block
    double* part; size_t part_batch, part_doubles;     // workspace, grown with the batch
    double* shares;
    std::mutex mu;
};

extern "C" int fmpc_est_create(fmpc_est* out, int len, int first, int d, int ndiv, const double* D_re, const double* D_im,
                               double scale, const double* A_s, const double* b_s, int p, int nx, int device) {
    if (!out || !D_re || !D_im || !A_s || !b_s) return FMPC_E_NULL;
    *out = nullptr;
    if (len < 64 || len % 64 != 0 || d < 1 || d > 32 || first < 0 || first + d > len || ndiv < 1 || ndiv > FE_MAXDIV || nx < 1 ||
        p != ndiv * d * d) return FMPC_E_DIM;
    if (hipSetDevice(device) != hipSuccess) return FMPC_E_HIP;
    fmpc_est_s* e = new (std::nothrow) fmpc_est_s();
    if (!e) return FMPC_E_ALLOC;
    e->device = device; e->len = len; e->d = d; e->first = first; e->ndiv = ndiv; e->nx = nx; e->p = p; e->scale = scale;
    e->pool = nullptr; e->qlist = nullptr; e->part = nullptr; e->part_batch = 0; e->part_doubles = 0; e->shares = nullptr;
    std::vector<double> G, Fimg;
    e->rank = fmpc_host_estimator_gain(A_s, p, nx, G);
    fmpc_host_estimator_dft_images(len, d, first, Fimg);
    for (double v : G) if (!std::isfinite(v)) { delete e; return FMPC_E_DIM; }
    const size_t npx = (size_t)len * len;
    std::vector<double> pool;
    auto push = [&](const double* v, size_t cnt) { const size_t o = pool.size(); pool.insert(pool.end(), v, v + cnt); return o; };
    e->oDre = push(D_re, ndiv * npx); e->oDim = push(D_im, ndiv * npx); e->oF = push(Fimg.data(), Fimg.size());
    e->oG = push(G.data(), G.size()); e->ob = push(b_s, p);
    if (hipMalloc((void**)&e->pool, pool.size() * sizeof(double)) != hipSuccess) { delete e; return FMPC_E_ALLOC; }
    if (hipMemcpy(e->pool, pool.data(), pool.size() * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) { (void)hipFree(e->pool); delete e; return FMPC_E_HIP; }
    // the range of k-steps of every row block that sees the pupil (fmpc_est_psf skips what lies outside)
    {
        const int nblk = len / 16, nks = len / 4;
        std::vector<int> qr((size_t)2 * nblk, 0);
        for (int b = 0; b < nblk; ++b) {
            int lo = nks, hi = 0;
            for (int Q = 0; Q < nks; ++Q) {
                bool any = false;
                for (int k = 0; k < ndiv && !any; ++k)
                    for (int x = 4 * Q; x < 4 * Q + 4 && !any; ++x)
                        for (int y = 16 * b; y < 16 * b + 16; ++y) {
                            const size_t o = (size_t)k * npx + (size_t)x * len + y;            // column-major: (row y, column x)
                            if (D_re[o] != 0.0 || D_im[o] != 0.0) { any = true; break; }
                        }
                if (any) { if (Q < lo) lo = Q; hi = Q + 1; }
            }
            if (hi <= lo) { lo = 0; hi = 0; }
            qr[2 * b] = lo; qr[2 * b + 1] = hi;
        }
        if (hipMalloc((void**)&e->qlist, qr.size() * sizeof(int)) != hipSuccess) { (void)hipFree(e->pool); delete e; return FMPC_E_ALLOC; }
        if (hipMemcpy(e->qlist, qr.data(), qr.size() * sizeof(int), hipMemcpyHostToDevice) != hipSuccess) { (void)hipFree(e->qlist); (void)hipFree(e->pool); delete e; return FMPC_E_HIP; }
    }
    *out = e;
    return FMPC_OK;
}

extern "C" int fmpc_est_destroy(fmpc_est e) {
    if (!e) return FMPC_E_NULL;
    (void)hipSetDevice(e->device);
    (void)hipDeviceSynchronize();
    if (e->pool) (void)hipFree(e->pool);
    if (e->qlist) (void)hipFree(e->qlist);
    if (e->part) (void)hipFree(e->part);
    if (e->shares) (void)hipFree(e->shares);
    delete e;
    return FMPC_OK;
}

extern "C" int fmpc_est_dims(fmpc_est e, int* len, int* d, int* ndiv, int* nx, int* p, int* rank) {
    if (!e) return FMPC_E_NULL;
    if (len) *len = e->len; if (d) *d = e->d; if (ndiv) *ndiv = e->ndiv; if (nx) *nx = e->nx; if (p) *p = e->p; if (rank) *rank = e->rank;
    return FMPC_OK;
}

extern "C" int fmpc_est_apply_device(fmpc_est e, int batch, const double* scrn, const double* noise, double* ad_est, double* Y_out,
                                     void* stream) {
    if (!e || !scrn || !ad_est) return FMPC_E_NULL;
    if (batch < 0) return FMPC_E_DIM;
    if (batch == 0) return FMPC_OK;
    if (hipSetDevice(e->device) != hipSuccess) return FMPC_E_HIP;
    std::lock_guard<std::mutex> lk(e->mu);
    if ((size_t)batch > e->part_batch) {
        (void)hipDeviceSynchronize();
        if (e->part) (void)hipFree(e->part);
        if (e->shares) (void)hipFree(e->shares);
            e->part = nullptr; e->shares = nullptr; e->part_batch = 0;
        size_t cap = 1;
        while (cap < (size_t)batch) cap *= 2;
        // (few screens: the PSF kernel may split the columns of a row block over two workgroups -- room for 4 screens x 128 partial windows per diversity)
        const size_t pw = cap * e->ndiv * (e->len / 16) < (size_t)4 * e->ndiv * 128 ? (size_t)4 * e->ndiv * 128 : cap * e->ndiv * (e->len / 16);
        e->part_doubles = pw * 2048;
        if (hipMalloc((void**)&e->part, pw * 2048 * sizeof(double)) != hipSuccess ||
            hipMalloc((void**)&e->shares, cap * e->ndiv * 4 * e->nx * sizeof(double)) != hipSuccess) return FMPC_E_ALLOC;
        e->part_batch = cap;
    }
    FeParams P;
    P.len = e->len; P.d = e->d; P.ndiv = e->ndiv; P.nx = e->nx; P.batch = batch; P.scale = e->scale;
    P.scrn = scrn; P.noise = noise; P.Dre = e->pool + e->oDre; P.Dim = e->pool + e->oDim; P.Fimg = e->pool + e->oF;
    P.qrange = e->qlist;
    P.G = e->pool + e->oG; P.bs = e->pool + e->ob; P.part = e->part; P.shares = e->shares; P.shares_cap = e->part_batch * (size_t)e->ndiv * 4 * e->nx; P.nshare = 1; P.part_cap = e->part_doubles; P.ad_est = ad_est; P.Yout = Y_out;
    return fmpc_launch_estimator(P, (hipStream_t)stream) == hipSuccess ? FMPC_OK : FMPC_E_HIP;
}

extern "C" int fmpc_est_apply(fmpc_est e, int batch, const double* scrn, const double* noise, double* ad_est, double* Y_out) {
    if (!e || !scrn || !ad_est) return FMPC_E_NULL;
    if (batch < 0) return FMPC_E_DIM;
    if (batch == 0) return FMPC_OK;
    if (hipSetDevice(e->device) != hipSuccess) return FMPC_E_HIP;
    const size_t npx = (size_t)e->len * e->len;
    double *ds = nullptr, *dn = nullptr, *da = nullptr, *dy = nullptr;
    int rc = FMPC_OK;
    if (hipMalloc((void**)&ds, batch * npx * sizeof(double)) != hipSuccess || hipMalloc((void**)&da, (size_t)batch * e->nx * sizeof(double)) != hipSuccess ||
        (noise && hipMalloc((void**)&dn, (size_t)batch * e->p * sizeof(double)) != hipSuccess) ||
        (Y_out && hipMalloc((void**)&dy, (size_t)batch * e->p * sizeof(double)) != hipSuccess)) rc = FMPC_E_ALLOC;
    if (rc == FMPC_OK && (hipMemcpy(ds, scrn, batch * npx * sizeof(double), hipMemcpyHostToDevice) != hipSuccess ||
                          (noise && hipMemcpy(dn, noise, (size_t)batch * e->p * sizeof(double), hipMemcpyHostToDevice) != hipSuccess))) rc = FMPC_E_HIP;
    if (rc == FMPC_OK) rc = fmpc_est_apply_device(e, batch, ds, dn, da, dy, nullptr);
    if (rc == FMPC_OK && (hipDeviceSynchronize() != hipSuccess || hipMemcpy(ad_est, da, (size_t)batch * e->nx * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess ||
                          (Y_out && hipMemcpy(Y_out, dy, (size_t)batch * e->p * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess))) rc = FMPC_E_HIP;
    if (ds) (void)hipFree(ds); if (dn) (void)hipFree(dn); if (da) (void)hipFree(da); if (dy) (void)hipFree(dy);
    return rc;
}
