// sr_traj.hip -- trajectory front end: raw coordinates -> unit X-H bond
// vectors, in the lab frame and after a least-squares superposition of every frame onto a reference structure.
//
// Reference semantics (calculate-Ct-from-traj.py):
//   obtain_XHvecs :64-86           vecXH = take(xyz, indexH) - take(xyz, indexX)   (float32, MDTraj coordinates)
//                                  vecXH = vecnorm_NDarray(vecXH, axis=2)          (transforms3d_supplement.py:40-52:
//                                          v / linalg.norm(v), nan_to_num: 0/0 -> 0, +-inf -> +-largest float32)
//   trj.center_coordinates(); trj.superpose(ref, frame=0, atom_indices=fit_indices)   :466-467 (MDTraj, third party,
//                                  absent from this image: unweighted optimal rotation of the fit atoms about their
//                                  centroid, applied to every atom) followed by obtain_XHvecs again.
// A bond vector is a coordinate DIFFERENCE, so translations drop out and the fitted vector is R_n (x_H - x_X) with R_n the
// optimal rotation of frame n.  R_n comes from Horn's closed form: the eigenvector of the largest eigenvalue of the
// symmetric 4 x 4 matrix built from S = sum_i (p_i - <p>)(r_i - <r>)^T is the unit quaternion that rotates the frame onto
// the reference (always a proper rotation).  The 4 x 4 problem is solved by cyclic Jacobi sweeps in registers.
//
// One workgroup per frame: the frame's coordinates are one contiguous block (nAtoms * 12 B), the fit atoms and the bond
// atoms are gathered from it by index (first touch pulls the lines into L1/L2; a frame is read from HBM once), the nine
// float64 sums are reduced by DPP wave sums + a fixed-order combine, every thread solves the same 4 x 4 problem, then the
// threads run over the bonds.  HBM-bound by construction: 12 B x atoms touched in, 24 B x bonds out per frame.
#include "sr_internal.h"

namespace {

struct XhArgs {
    const float *xyz;        // (nFrames, nAtoms, 3)
    int64_t nFrames, nAtoms;
    const int *idxX, *idxH;  // (nV) device
    int nV;
    const int *fit;          // (nFit) device or null
    const double *refc;      // (nFit, 3) reference positions of the fit atoms, centred on their centroid
    int nFit;
    float *lab;              // (nFrames, nV, 3) or null
    float *fitted;           // (nFrames, nV, 3) or null
    double *quat;            // (nFrames, 4) or null
};

template <int P, int Q>
__device__ __forceinline__ void jacobi_rot(double (&A)[4][4], double (&V)[4][4])
{
    const double apq = A[P][Q];
    if (fabs(apq) < 1e-300) return;
    const double theta = (A[Q][Q] - A[P][P]) / (2.0 * apq);
    const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
    const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
    const double app = A[P][P], aqq = A[Q][Q];
    A[P][P] = app - t * apq;
    A[Q][Q] = aqq + t * apq;
    A[P][Q] = A[Q][P] = 0.0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (k != P && k != Q) {
            const double akp = A[k][P], akq = A[k][Q];
            A[k][P] = A[P][k] = c * akp - s * akq;
            A[k][Q] = A[Q][k] = s * akp + c * akq;
        }
        const double vkp = V[k][P], vkq = V[k][Q];
        V[k][P] = c * vkp - s * vkq;
        V[k][Q] = s * vkp + c * vkq;
    }
}

// unit quaternion (w, x, y, z) of the rotation R that minimises sum |R p_i - r_i|^2, from S[a][b] = sum p_a r_b (centred)
__device__ __forceinline__ void horn_quaternion(const double *S, double *q)
{
    const double Sxx = S[0], Sxy = S[1], Sxz = S[2], Syx = S[3], Syy = S[4], Syz = S[5], Szx = S[6], Szy = S[7], Szz = S[8];
    double A[4][4] = {{Sxx + Syy + Szz, Syz - Szy, Szx - Sxz, Sxy - Syx},
                      {Syz - Szy, Sxx - Syy - Szz, Sxy + Syx, Szx + Sxz},
                      {Szx - Sxz, Sxy + Syx, -Sxx + Syy - Szz, Syz + Szy},
                      {Sxy - Syx, Szx + Sxz, Syz + Szy, -Sxx - Syy + Szz}};
    double V[4][4] = {{1, 0, 0, 0}, {0, 1, 0, 0}, {0, 0, 1, 0}, {0, 0, 0, 1}};
    for (int sweep = 0; sweep < 12; ++sweep) {
        const double off = fabs(A[0][1]) + fabs(A[0][2]) + fabs(A[0][3]) + fabs(A[1][2]) + fabs(A[1][3]) + fabs(A[2][3]);
        const double dia = fabs(A[0][0]) + fabs(A[1][1]) + fabs(A[2][2]) + fabs(A[3][3]);
        if (off <= 1e-17 * dia) break;
        jacobi_rot<0, 1>(A, V);
        jacobi_rot<0, 2>(A, V);
        jacobi_rot<0, 3>(A, V);
        jacobi_rot<1, 2>(A, V);
        jacobi_rot<1, 3>(A, V);
        jacobi_rot<2, 3>(A, V);
    }
    int best = 0;
    double lam = A[0][0];
#pragma unroll
    for (int k = 1; k < 4; ++k)
        if (A[k][k] > lam) { lam = A[k][k]; best = k; }
    double w = 0, x = 0, y = 0, z = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (k == best) { w = V[0][k]; x = V[1][k]; y = V[2][k]; z = V[3][k]; }
    const double n = 1.0 / sqrt((w * w + x * x) + (y * y + z * z));
    const double sg = w < 0 ? -n : n;
    q[0] = w * sg; q[1] = x * sg; q[2] = y * sg; q[3] = z * sg;
}

__device__ __forceinline__ float nan_to_num_f32(float v)
{
    // numpy.nan_to_num defaults: nan -> 0, +-inf -> +-largest finite
    if (v != v) return 0.f;
    if (v > 3.4028234663852886e38f) return 3.4028234663852886e38f;
    if (v < -3.4028234663852886e38f) return -3.4028234663852886e38f;
    return v;
}

__global__ __launch_bounds__(256) void k_xh_vectors(XhArgs a)
{
#pragma clang fp contract(off)
    __shared__ double red[4 * 9];
    __shared__ double Ssh[9];
    const int tid = threadIdx.x;
    const int64_t n = blockIdx.x;
    const float *fr = a.xyz + n * a.nAtoms * 3;
    double R[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
    const bool do_fit = a.fit != nullptr && (a.fitted != nullptr || a.quat != nullptr);
    if (do_fit) {
        double acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        for (int i = tid; i < a.nFit; i += 256) {
            const float *p = fr + (int64_t)a.fit[i] * 3;
            const double px = (double)p[0], py = (double)p[1], pz = (double)p[2];
            const double rx = a.refc[i * 3 + 0], ry = a.refc[i * 3 + 1], rz = a.refc[i * 3 + 2];
            // the reference positions are centred (sum r = 0), so sum (p - <p>) r^T = sum p r^T
            acc[0] = fma(px, rx, acc[0]); acc[1] = fma(px, ry, acc[1]); acc[2] = fma(px, rz, acc[2]);
            acc[3] = fma(py, rx, acc[3]); acc[4] = fma(py, ry, acc[4]); acc[5] = fma(py, rz, acc[5]);
            acc[6] = fma(pz, rx, acc[6]); acc[7] = fma(pz, ry, acc[7]); acc[8] = fma(pz, rz, acc[8]);
        }
        const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
        for (int m = 0; m < 9; ++m) {
            const double t = sr_wave_sum_f64(acc[m]);
            if (lane == 0) red[wave * 9 + m] = t;
        }
        __syncthreads();
        if (tid < 9) Ssh[tid] = ((red[tid] + red[9 + tid]) + red[18 + tid]) + red[27 + tid];
        __syncthreads();
        double S[9], q[4];
#pragma unroll
        for (int m = 0; m < 9; ++m) S[m] = Ssh[m];
        horn_quaternion(S, q);
        if (a.quat && tid < 4) {
            double v = q[0];
            if (tid == 1) v = q[1];
            if (tid == 2) v = q[2];
            if (tid == 3) v = q[3];
            a.quat[n * 4 + tid] = v;
        }
        const double w = q[0], x = q[1], y = q[2], z = q[3];
        R[0][0] = 1 - 2 * (y * y + z * z); R[0][1] = 2 * (x * y - w * z);     R[0][2] = 2 * (x * z + w * y);
        R[1][0] = 2 * (x * y + w * z);     R[1][1] = 1 - 2 * (x * x + z * z); R[1][2] = 2 * (y * z - w * x);
        R[2][0] = 2 * (x * z - w * y);     R[2][1] = 2 * (y * z + w * x);     R[2][2] = 1 - 2 * (x * x + y * y);
    }
    for (int v = tid; v < a.nV; v += 256) {
        const float *pH = fr + (int64_t)a.idxH[v] * 3;
        const float *pX = fr + (int64_t)a.idxX[v] * 3;
        const float dx = pH[0] - pX[0], dy = pH[1] - pX[1], dz = pH[2] - pX[2];          // float32, like numpy on MDTraj's xyz
        const int64_t o = (n * a.nV + v) * 3;
        if (a.lab) {
            const float nr = sqrtf((dx * dx + dy * dy) + dz * dz);                        // linalg.norm: sqrt(add.reduce(x*x))
            a.lab[o + 0] = nan_to_num_f32(dx / nr);
            a.lab[o + 1] = nan_to_num_f32(dy / nr);
            a.lab[o + 2] = nan_to_num_f32(dz / nr);
        }
        if (a.fitted) {
            const double x = (double)dx, y = (double)dy, z = (double)dz;
            const double rx = (R[0][0] * x + R[0][1] * y) + R[0][2] * z;
            const double ry = (R[1][0] * x + R[1][1] * y) + R[1][2] * z;
            const double rz = (R[2][0] * x + R[2][1] * y) + R[2][2] * z;
            const double nr = sqrt((rx * rx + ry * ry) + rz * rz);
            a.fitted[o + 0] = nan_to_num_f32((float)(rx / nr));
            a.fitted[o + 1] = nan_to_num_f32((float)(ry / nr));
            a.fitted[o + 2] = nan_to_num_f32((float)(rz / nr));
        }
    }
}

}  // namespace

extern "C" {

int sr_xh_vectors_f32_dev(sr_ctx *ctx, const float *xyz, int64_t nFrames, int64_t nAtoms, const int32_t *idxX_host,
                          const int32_t *idxH_host, int nV, const int32_t *fit_idx_host, int nFit, const float *ref_xyz_host,
                          float *vec_lab, float *vec_fit, double *quat)
{
    SR_CHECK_CTX(ctx);
    SR_REQUIRE(xyz && idxX_host && idxH_host, -2, "sr_xh_vectors_f32_dev: null pointer");
    SR_REQUIRE(nFrames >= 1 && nAtoms >= 1 && nV >= 1 && nFrames < ((int64_t)1 << 31), -3,
               "sr_xh_vectors_f32_dev: bad sizes nFrames=%lld nAtoms=%lld nV=%d", (long long)nFrames, (long long)nAtoms, nV);
    SR_REQUIRE(vec_lab || vec_fit || quat, -2, "sr_xh_vectors_f32_dev: no output requested");
    const bool fit = vec_fit != nullptr || quat != nullptr;
    if (fit) SR_REQUIRE(fit_idx_host && ref_xyz_host && nFit >= 3, -3, "sr_xh_vectors_f32_dev: a superposition needs >= 3 fit atoms and a reference structure");
    for (int v = 0; v < nV; ++v)
        SR_REQUIRE(idxX_host[v] >= 0 && idxX_host[v] < nAtoms && idxH_host[v] >= 0 && idxH_host[v] < nAtoms, -3,
                   "sr_xh_vectors_f32_dev: bond %d atom index out of range", v);
    // every host table is checked BEFORE anything is queued, and the (small) table copies are complete when this function
    // returns: the caller may free or reuse idxX / idxH / fit_idx / ref_xyz right away, pinned or not
    if (fit)
        for (int i = 0; i < nFit; ++i)
            SR_REQUIRE(fit_idx_host[i] >= 0 && fit_idx_host[i] < nAtoms, -3, "sr_xh_vectors_f32_dev: fit atom %d out of range", i);
    const size_t ibytes = (size_t)(2 * nV + (fit ? nFit : 0)) * sizeof(int);
    int *idx_d = (int *)sr_workspace(ctx, SR_WS_IN2, ibytes);
    double *refc_d = fit ? (double *)sr_workspace(ctx, SR_WS_IN1, (size_t)nFit * 3 * sizeof(double)) : nullptr;
    if (!idx_d || (fit && !refc_d)) return -5;
    double *tmp = nullptr;
    hipError_t e = hipMemcpyAsync(idx_d, idxX_host, (size_t)nV * sizeof(int), hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(idx_d + nV, idxH_host, (size_t)nV * sizeof(int), hipMemcpyHostToDevice, ctx->stream);
    if (fit && e == hipSuccess) {
        double c[3] = {0, 0, 0};
        for (int i = 0; i < nFit; ++i)
            for (int k = 0; k < 3; ++k) c[k] += (double)ref_xyz_host[(size_t)fit_idx_host[i] * 3 + k];
        for (int k = 0; k < 3; ++k) c[k] /= (double)nFit;
        tmp = new double[(size_t)nFit * 3];
        for (int i = 0; i < nFit; ++i)
            for (int k = 0; k < 3; ++k) tmp[(size_t)i * 3 + k] = (double)ref_xyz_host[(size_t)fit_idx_host[i] * 3 + k] - c[k];
        e = hipMemcpyAsync(refc_d, tmp, (size_t)nFit * 3 * sizeof(double), hipMemcpyHostToDevice, ctx->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(idx_d + 2 * nV, fit_idx_host, (size_t)nFit * sizeof(int), hipMemcpyHostToDevice, ctx->stream);
    }
    const hipError_t es = hipStreamSynchronize(ctx->stream);
    delete[] tmp;
    SR_HIP(e);
    SR_HIP(es);
    XhArgs a;
    a.xyz = xyz; a.nFrames = nFrames; a.nAtoms = nAtoms; a.idxX = idx_d; a.idxH = idx_d + nV; a.nV = nV;
    a.fit = fit ? idx_d + 2 * nV : nullptr; a.refc = refc_d; a.nFit = fit ? nFit : 0;
    a.lab = vec_lab; a.fitted = vec_fit; a.quat = quat;
    hipLaunchKernelGGL(k_xh_vectors, dim3((unsigned)nFrames), dim3(256), 0, ctx->stream, a);
    SR_HIP(hipGetLastError());
    return 0;
}

int sr_xh_vectors_f32(sr_ctx *ctx, const float *xyz, int64_t nFrames, int64_t nAtoms, const int32_t *idxX, const int32_t *idxH,
                      int nV, const int32_t *fit_idx, int nFit, const float *ref_xyz, float *vec_lab, float *vec_fit, double *quat)
{
    SR_CHECK_CTX(ctx);
    SR_REQUIRE(xyz && idxX && idxH, -2, "sr_xh_vectors_f32: null pointer");
    SR_REQUIRE(nFrames >= 1 && nAtoms >= 1 && nV >= 1, -3, "sr_xh_vectors_f32: bad sizes");
    const size_t in_bytes = (size_t)nFrames * nAtoms * 3 * sizeof(float);
    const size_t vb = (size_t)nFrames * nV * 3 * sizeof(float);
    float *xyz_d = (float *)sr_workspace(ctx, SR_WS_VECS, in_bytes);
    float *lab_d = vec_lab ? (float *)sr_workspace(ctx, SR_WS_OUT0, vb) : nullptr;
    float *fit_d = vec_fit ? (float *)sr_workspace(ctx, SR_WS_OUT1, vb) : nullptr;
    double *q_d = quat ? (double *)sr_workspace(ctx, SR_WS_OUT2, (size_t)nFrames * 4 * sizeof(double)) : nullptr;
    if (!xyz_d || (vec_lab && !lab_d) || (vec_fit && !fit_d) || (quat && !q_d)) return -5;
    SR_HIP(hipMemcpyAsync(xyz_d, xyz, in_bytes, hipMemcpyHostToDevice, ctx->stream));
    int rc = sr_xh_vectors_f32_dev(ctx, xyz_d, nFrames, nAtoms, idxX, idxH, nV, fit_idx, nFit, ref_xyz, lab_d, fit_d, q_d);
    if (rc) return rc;
    if (vec_lab) SR_HIP(hipMemcpyAsync(vec_lab, lab_d, vb, hipMemcpyDeviceToHost, ctx->stream));
    if (vec_fit) SR_HIP(hipMemcpyAsync(vec_fit, fit_d, vb, hipMemcpyDeviceToHost, ctx->stream));
    if (quat) SR_HIP(hipMemcpyAsync(quat, q_d, (size_t)nFrames * 4 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    SR_HIP(hipStreamSynchronize(ctx->stream));
    return 0;
}

}  // extern "C"
