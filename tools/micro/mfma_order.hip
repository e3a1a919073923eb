// mfma_order.hip -- does a float64 MFMA accumulate its k products as one FMA chain in ascending k, starting
// from C?  (The stacking kernels are bit-exact FMA chains in a fixed order; an MFMA form of the multilinear
// blend, DESIGN "what comes next", needs the same result.)  Also discovers the operand layout of
// v_mfma_f64_4x4x4 (4 blocks) by basis probes.   hipcc --offload-arch=gfx950 -O2 mfma_order.hip -o mfma_order
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef double v4f64 __attribute__((ext_vector_type(4)));

__global__ void k16(const double *A, const double *B, const double *C, double *D)
{
    const int l = threadIdx.x;
    // A lane l -> A[i = l & 15][k = l >> 4]; B lane l -> B[k = l >> 4][j = l & 15]; D reg r -> row (l >> 4) + 4 r, col l & 15
    const double a = A[(l & 15) * 4 + (l >> 4)], b = B[(l >> 4) * 16 + (l & 15)];
    v4f64 c;
    for (int r = 0; r < 4; r++) c[r] = C[((l >> 4) + 4 * r) * 16 + (l & 15)];
    c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; r++) D[((l >> 4) + 4 * r) * 16 + (l & 15)] = c[r];
}

__global__ void k4(const double *a_in, const double *b_in, const double *c_in, double *d_out, int n)
{
    const int l = threadIdx.x;
    for (int t = 0; t < n; t++)
        d_out[t * 64 + l] = __builtin_amdgcn_mfma_f64_4x4x4f64(a_in[t * 64 + l], b_in[t * 64 + l], c_in[t * 64 + l], 0, 0, 0);
}

static double chain(const double *a, const double *b, double c, int order)
{
    double acc = c;
    if (order == 0) for (int k = 0; k < 4; k++) acc = fma(a[k], b[k], acc);
    else if (order == 1) for (int k = 3; k >= 0; k--) acc = fma(a[k], b[k], acc);
    else { double s = 0.0; for (int k = 0; k < 4; k++) s = fma(a[k], b[k], s); acc = c + s; }
    return acc;
}

int main()
{
    srand(7);
    auto rnd = []() { return (rand() / (double)RAND_MAX - 0.5) * exp((rand() % 9) - 4.0); };
    // ---- 16x16x4
    std::vector<double> A(64), B(64), C(256), D(256);
    double *dA, *dB, *dC, *dD;
    hipMalloc(&dA, 64 * 8); hipMalloc(&dB, 64 * 8); hipMalloc(&dC, 256 * 8); hipMalloc(&dD, 256 * 8);
    long eq[3] = {0, 0, 0}, tot = 0;
    for (int rep = 0; rep < 200; rep++) {
        for (auto &x : A) x = rnd();
        for (auto &x : B) x = rnd();
        for (auto &x : C) x = rnd();
        hipMemcpy(dA, A.data(), 64 * 8, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), 64 * 8, hipMemcpyHostToDevice);
        hipMemcpy(dC, C.data(), 256 * 8, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k16, dim3(1), dim3(64), 0, 0, dA, dB, dC, dD);
        hipMemcpy(D.data(), dD, 256 * 8, hipMemcpyDeviceToHost);
        for (int i = 0; i < 16; i++) for (int j = 0; j < 16; j++) {
            double a[4], b[4];
            for (int k = 0; k < 4; k++) { a[k] = A[i * 4 + k]; b[k] = B[k * 16 + j]; }
            for (int o = 0; o < 3; o++) eq[o] += memcmp(&D[i * 16 + j], (double[]){chain(a, b, C[i * 16 + j], o)}, 8) == 0;
            tot++;
        }
    }
    printf("16x16x4: of %ld outputs bitwise equal to  ascending FMA chain from C: %ld   descending: %ld   sum-then-add: %ld\n",
           tot, eq[0], eq[1], eq[2]);
    // ---- 4x4x4 (4 blocks): layout by basis probes: A = e_p, B = 1 -> which D lanes see A lane p; B = e_q, A = 1 likewise
    const int NP = 128;
    std::vector<double> a(NP * 64, 0.0), b(NP * 64, 0.0), c(NP * 64, 0.0), d(NP * 64);
    for (int p = 0; p < 64; p++) { for (int l = 0; l < 64; l++) b[p * 64 + l] = 1.0; a[p * 64 + p] = 1.0; }
    for (int q = 0; q < 64; q++) { for (int l = 0; l < 64; l++) a[(64 + q) * 64 + l] = 1.0; b[(64 + q) * 64 + q] = 1.0; }
    double *da, *db, *dc, *dd;
    hipMalloc(&da, NP * 64 * 8); hipMalloc(&db, NP * 64 * 8); hipMalloc(&dc, NP * 64 * 8); hipMalloc(&dd, NP * 64 * 8);
    hipMemcpy(da, a.data(), NP * 64 * 8, hipMemcpyHostToDevice); hipMemcpy(db, b.data(), NP * 64 * 8, hipMemcpyHostToDevice);
    hipMemcpy(dc, c.data(), NP * 64 * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k4, dim3(1), dim3(64), 0, 0, da, db, dc, dd, NP);
    hipMemcpy(d.data(), dd, NP * 64 * 8, hipMemcpyDeviceToHost);
    // usesA[lane of D][lane of A], usesB[...]
    static int uA[64][64], uB[64][64];
    for (int p = 0; p < 64; p++) for (int l = 0; l < 64; l++) { uA[l][p] = d[p * 64 + l] != 0.0; uB[l][p] = d[(64 + p) * 64 + l] != 0.0; }
    printf("4x4x4: D lane 0 uses A lanes:"); for (int p = 0; p < 64; p++) if (uA[0][p]) printf(" %d", p);
    printf("   B lanes:"); for (int p = 0; p < 64; p++) if (uB[0][p]) printf(" %d", p);
    printf("\n4x4x4: D lane 5 uses A lanes:"); for (int p = 0; p < 64; p++) if (uA[5][p]) printf(" %d", p);
    printf("   B lanes:"); for (int p = 0; p < 64; p++) if (uB[5][p]) printf(" %d", p);
    printf("\n4x4x4: D lane 21 uses A lanes:"); for (int p = 0; p < 64; p++) if (uA[21][p]) printf(" %d", p);
    printf("   B lanes:"); for (int p = 0; p < 64; p++) if (uB[21][p]) printf(" %d", p);
    printf("\n");
    // order test with the discovered pairing: for D lane l the k-th A lane pairs with the k-th B lane (ascending lane order)
    long e4[3] = {0, 0, 0}, t4 = 0;
    for (int rep = 0; rep < 100; rep++) {
        std::vector<double> ra(64), rb(64), rc(64), rd(64);
        for (auto &x : ra) x = rnd();
        for (auto &x : rb) x = rnd();
        for (auto &x : rc) x = rnd();
        hipMemcpy(da, ra.data(), 64 * 8, hipMemcpyHostToDevice); hipMemcpy(db, rb.data(), 64 * 8, hipMemcpyHostToDevice);
        hipMemcpy(dc, rc.data(), 64 * 8, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k4, dim3(1), dim3(64), 0, 0, da, db, dc, dd, 1);
        hipMemcpy(rd.data(), dd, 64 * 8, hipMemcpyDeviceToHost);
        for (int l = 0; l < 64; l++) {
            double aa[4], bb[4];
            int na = 0, nb = 0;
            for (int p = 0; p < 64 && na < 4; p++) if (uA[l][p]) aa[na++] = ra[p];
            for (int p = 0; p < 64 && nb < 4; p++) if (uB[l][p]) bb[nb++] = rb[p];
            if (na != 4 || nb != 4) continue;
            for (int o = 0; o < 3; o++) { double r = chain(aa, bb, rc[l], o); e4[o] += memcmp(&rd[l], &r, 8) == 0; }
            t4++;
        }
    }
    printf("4x4x4: of %ld outputs bitwise equal to  ascending chain (A/B lanes paired in lane order): %ld   descending: %ld   sum-then-add: %ld\n",
           t4, e4[0], e4[1], e4[2]);
    return 0;
}
