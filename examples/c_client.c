/* A plain-C client of the drop-in boundary (include/fastmpc.h): what a MEX gateway or any compiled host of the reference
 * would do per timestep -- the arguments of Fast_MPC2(...) (VAR_2/Fast_MPC2.m:28-29) and mpc_fixed_log_newton(nw, k)
 * (:124-130) as column-major fp64 arrays, one call to fmpc_solve_once, x_opt back.
 *
 *   gcc -O2 -Iinclude examples/c_client.c -o examples/c_client -Lmpc-sensorlessao_amd/lib -lfastmpc -Wl,-rpath,$PWD/mpc-sensorlessao_amd/lib -lm
 *   examples/c_client model.bin out.bin     (model.bin: written by tests/test_gpu_c_client.py)
 *
 * File format (all little-endian): int32 n, m, T, nw; double k; then Q (n*n) R (m*m) Qf (n*n) x_min x_max (n) u_min u_max (m)
 * x0 x0_pre (n) A1 A2 (n*n) B (n*m) w (T*n) nu0 (T*n), every matrix column-major.  Output: int32 rc, iters; double z[T*(n+m)].
 */
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include "fastmpc.h"

static double* rd(FILE* f, size_t cnt) {
    double* p = (double*)malloc(cnt * sizeof(double));
    if (!p || fread(p, sizeof(double), cnt, f) != cnt) { fprintf(stderr, "short read\n"); exit(2); }
    return p;
}

int main(int argc, char** argv) {
    if (argc < 3) { fprintf(stderr, "usage: c_client model.bin out.bin\n"); return 2; }
    FILE* f = fopen(argv[1], "rb");
    if (!f) { perror("model"); return 2; }
    int32_t hd[4]; double k;
    if (fread(hd, sizeof(int32_t), 4, f) != 4 || fread(&k, sizeof(double), 1, f) != 1) return 2;
    const int n = hd[0], m = hd[1], T = hd[2], nw = hd[3];
    double* Q = rd(f, (size_t)n * n); double* R = rd(f, (size_t)m * m); double* Qf = rd(f, (size_t)n * n);
    double* xmin = rd(f, n); double* xmax = rd(f, n); double* umin = rd(f, m); double* umax = rd(f, m);
    double* x0 = rd(f, n); double* x0p = rd(f, n);
    double* A1 = rd(f, (size_t)n * n); double* A2 = rd(f, (size_t)n * n); double* B = rd(f, (size_t)n * m);
    double* w = rd(f, (size_t)T * n); double* nu0 = rd(f, (size_t)T * n);
    fclose(f);
    const size_t nz = (size_t)T * (n + m);
    double* z = (double*)calloc(nz, sizeof(double));
    int iters = -1, rc = 0;
    /* the reference rebuilds its object every timestep with the same model (README.md:548): three calls, the last one is reported
       (the library keeps the handle of the model between the calls) */
    for (int rep = 0; rep < 3; ++rep)
        rc = fmpc_solve_once(n, m, T, 2, Q, R, NULL, Qf, NULL, NULL, NULL, xmin, xmax, umin, umax, NULL, NULL,
                             x0, x0p, NULL, A1, A2, B, w, NULL, NULL, nu0, nw, k, 0, z, &iters);
    if (rc < 0) fprintf(stderr, "fmpc_solve_once: %s\n", fmpc_strerror(rc));
    fmpc_solve_once_cache_clear();
    FILE* o = fopen(argv[2], "wb");
    if (!o) { perror("out"); return 2; }
    int32_t res[2] = {rc, iters};
    fwrite(res, sizeof(int32_t), 2, o);
    fwrite(z, sizeof(double), nz, o);
    fclose(o);
    return rc < 0 ? 1 : 0;
}
