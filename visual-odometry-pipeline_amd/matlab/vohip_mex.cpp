// vohip_mex.cpp -- MEX gateway between MATLAB and the C ABI of libvohip.so (include/vohip.h).
//
// Build (on a machine with MATLAB and ROCm; neither mex.h nor MATLAB exist in the container
// this library is developed in, so this file is compiled and exercised only there):
//     mex -R2018a vohip_mex.cpp -I<repo>/include -L<repo>/visual-odometry-pipeline_amd/lib -lvohip
//
// Usage from the .m wrappers in this directory:  [out...] = vohip_mex('opcode', args...)
// One persistent context per MATLAB session (device 0 unless VOHIP_DEVICE is set); it is
// destroyed by mexAtExit.  Every array crosses the boundary exactly as MATLAB stores it
// (column-major), which is the layout the C ABI is specified in -- no transposes.
// MATLAB calls MEX functions on its interpreter thread only; the context is never shared.
// Errors: the C ABI returns status codes; they are turned into mexErrMsgIdAndTxt here.  That call
// long-jumps out of the gateway, so no object with a destructor lives in this file: temporaries are
// mxCalloc blocks, which MATLAB releases itself when a MEX function ends with an error.
// tests/test_mex_gateway.py compiles this file against a declaration-only mex.h, so a signature that
// drifts away from vohip.h breaks the build.
#include <cstdint>
#include <cstdlib>
#include <cstring>

#include "mex.h"
#include "vohip.h"

static vohip_ctx *g_ctx = nullptr;
static vohip_frame *g_frame = nullptr;   // the device-resident loop state ('frame_*' opcodes)
static vohip_window *g_window = nullptr; // the bundle-adjustment window ('window_*' opcodes)
static vohip_group *g_group = nullptr;   // a group of loops advanced in lockstep ('group_*' opcodes)
static int g_group_n = 0;                // its number of sequences

static void at_exit()
{
    if (g_window) vohip_window_destroy(g_window);
    g_window = nullptr;
    if (g_group) vohip_group_destroy(g_group);
    g_group = nullptr;
    if (g_frame) vohip_frame_destroy(g_frame);
    g_frame = nullptr;
    if (g_ctx) vohip_destroy(g_ctx);
    g_ctx = nullptr;
}

static vohip_ctx *ctx()
{
    if (!g_ctx) {
        const char *dev = std::getenv("VOHIP_DEVICE");
        if (vohip_create(&g_ctx, dev ? std::atoi(dev) : 0, nullptr) != VOHIP_OK)
            mexErrMsgIdAndTxt("vohip:create", "%s", vohip_last_error(nullptr));
        mexAtExit(at_exit);
        mexLock();
    }
    return g_ctx;
}

static void check(int status, bool allow_no_model = false)
{
    if (status == VOHIP_OK || (allow_no_model && status == VOHIP_NO_MODEL)) return;
    mexErrMsgIdAndTxt("vohip:status", "libvohip status %d: %s", status, vohip_last_error(g_ctx));
}

static const double *dbl(const mxArray *a, const char *what)
{
    if (!mxIsDouble(a) || mxIsComplex(a)) mexErrMsgIdAndTxt("vohip:type", "%s must be real double", what);
    return mxGetDoubles(a);
}
static int numel(const mxArray *a) { return (int)mxGetNumberOfElements(a); }
static int cols(const mxArray *a) { return (int)mxGetN(a); }
static int rows(const mxArray *a) { return (int)mxGetM(a); }

template <class T>
static T *scratch(size_t n)   // zeroed, owned by MATLAB: freed by mxFree or, after an error, by MATLAB
{
    return static_cast<T *>(mxCalloc(n ? n : 1, sizeof(T)));
}

// 0-based int32 copy of a 1-based MATLAB index matrix
static int32_t *idx0(const mxArray *a)
{
    const double *p = dbl(a, "index matrix");
    const size_t n = (size_t)numel(a);
    int32_t *v = scratch<int32_t>(n);
    for (size_t i = 0; i < n; ++i) v[i] = (int32_t)p[i] - 1;
    return v;
}

// n x 2 MATLAB matrix ([x...; y...] in memory) -> 2 x n, the layout the library takes points in
static double *interleave2(const double *p, int n)
{
    double *a = scratch<double>((size_t)2 * n);
    for (int i = 0; i < n; ++i) {
        a[2 * i] = p[i];
        a[2 * i + 1] = p[n + i];
    }
    return a;
}

// [T_W_C, info, status] of the 'frame_step' / 'frame_retry' opcodes
static void frame_outputs(int nlhs, mxArray *plhs[], const vohip_frame_pose &pose, const vohip_frame_counts &counts,
                          int status)
{
    plhs[0] = mxCreateDoubleMatrix(4, 4, mxREAL);
    std::memcpy(mxGetDoubles(plhs[0]), pose.T_W_C, sizeof(double) * 16);
    if (nlhs > 1) {
        const char *names[] = {"n_tracked", "n_inliers", "n_iter", "num_iterations", "n_landmarks", "n_candidates",
                               "num_new", "n_detected", "n_kept"};
        const double values[] = {(double)pose.n_tracked, (double)pose.n_inliers, (double)pose.n_iter, pose.num_iterations,
                                 (double)counts.n_landmarks, (double)counts.n_candidates, (double)counts.num_new,
                                 (double)counts.n_detected, (double)counts.n_kept};
        plhs[1] = mxCreateStructMatrix(1, 1, 9, names);
        for (int k = 0; k < 9; ++k) mxSetField(plhs[1], 0, names[k], mxCreateDoubleScalar(values[k]));
    }
    if (nlhs > 2) plhs[2] = mxCreateDoubleScalar(status);
}

// fields of a 'frame_create' / 'group_create' parameter struct over the library's defaults
static void frame_params_from_struct(const mxArray *st, vohip_frame_params &p)
{
    auto field = [&](const char *name, double dflt) {
        const mxArray *f = mxGetField(st, 0, name);
        return f ? mxGetScalar(f) : dflt;
    };
    p.r_discard_redundant = field("r_discard_redundant", p.r_discard_redundant);
    p.angle_threshold_deg = field("angle_threshold_deg", p.angle_threshold_deg);
    p.min_depth = field("min_depth", p.min_depth);
    p.max_depth = field("max_depth", p.max_depth);
    p.min_quality = field("min_quality", p.min_quality);
    p.filter_size = (int)field("filter_size", p.filter_size);
    p.detector = (int)field("detector", p.detector);
    p.refine_pose = (int)field("refine_pose", p.refine_pose);
    p.is_BA = (int)field("is_BA", p.is_BA);
    p.min_contrast = field("min_contrast", p.min_contrast);
    p.metric_threshold = field("metric_threshold", p.metric_threshold);
    p.brisk_octaves = (int)field("brisk_octaves", p.brisk_octaves);
}

// info struct array (1 x n) of the 'group_step' opcode, same fields as frame_outputs'
static mxArray *group_info(const vohip_frame_pose *pose, const vohip_frame_counts *counts, int n)
{
    const char *names[] = {"n_tracked", "n_inliers", "n_iter", "num_iterations", "n_landmarks", "n_candidates",
                           "num_new", "n_detected", "n_kept"};
    mxArray *info = mxCreateStructMatrix(1, n, 9, names);
    for (int s = 0; s < n; ++s) {
        const double values[] = {(double)pose[s].n_tracked, (double)pose[s].n_inliers, (double)pose[s].n_iter,
                                 pose[s].num_iterations, (double)counts[s].n_landmarks, (double)counts[s].n_candidates,
                                 (double)counts[s].num_new, (double)counts[s].n_detected, (double)counts[s].n_kept};
        for (int k = 0; k < 9; ++k) mxSetField(info, s, names[k], mxCreateDoubleScalar(values[k]));
    }
    return info;
}

void mexFunction(int nlhs, mxArray *plhs[], int nrhs, const mxArray *prhs[])
{
    if (nrhs < 1 || !mxIsChar(prhs[0])) mexErrMsgIdAndTxt("vohip:usage", "vohip_mex('opcode', ...)");
    char opbuf[64];
    mxGetString(prhs[0], opbuf, sizeof(opbuf));
    const mxArray **in = prhs + 1;
    vohip_ctx *c = ctx();
    auto is = [&](const char *name) { return std::strcmp(opbuf, name) == 0; };

    if (is("ransac_p3p") || is("ransac_dlt")) {
        // [R, t, mask, max_hist, iter_hist, n_iter, status] = vohip_mex('ransac_p3p', kp_rc 2xN, X 3xN, K, samples 3xH (1-based))
        // status: 0 = a model, VOHIP_NO_MODEL (3) = none reached 30 inliers, VOHIP_SAMPLES_EXHAUSTED (4) = the
        // adaptive rule wants more than the H triples supplied (call again with more, the first H unchanged)
        // 'ransac_dlt': the use_p3p = false branch (ransacLocalization.m:20-24,69-74,145-151), samples 6xH
        const bool dlt = is("ransac_dlt");
        if (rows(in[3]) != (dlt ? 6 : 3)) mexErrMsgIdAndTxt("vohip:size", "samples must hold 3 (P3P) or 6 (DLT) rows");
        const int n = cols(in[0]), h = cols(in[3]);
        int32_t *s = idx0(in[3]);
        plhs[0] = mxCreateDoubleMatrix(3, 3, mxREAL);
        plhs[1] = mxCreateDoubleMatrix(3, 1, mxREAL);
        mxArray *mask = mxCreateLogicalMatrix(1, n);
        double *mh = scratch<double>((size_t)h), *ih = scratch<double>((size_t)h);
        int n_iter = 0;
        const int st = (dlt ? vohip_ransac_dlt : vohip_ransac_p3p)(c, dbl(in[0], "keypoints"), dbl(in[1], "landmarks"), n,
                                                                    dbl(in[2], "K"), s, h, mxGetDoubles(plhs[0]),
                                                                    mxGetDoubles(plhs[1]), (uint8_t *)mxGetLogicals(mask),
                                                                    mh, ih, &n_iter);
        if (st != VOHIP_SAMPLES_EXHAUSTED) check(st, true);
        plhs[2] = mask;
        plhs[3] = mxCreateDoubleMatrix(1, n_iter, mxREAL);
        plhs[4] = mxCreateDoubleMatrix(1, n_iter, mxREAL);
        std::memcpy(mxGetDoubles(plhs[3]), mh, sizeof(double) * n_iter);
        std::memcpy(mxGetDoubles(plhs[4]), ih, sizeof(double) * n_iter);
        plhs[5] = mxCreateDoubleScalar(n_iter);
        plhs[6] = mxCreateDoubleScalar(st);
        mxFree(s);
        mxFree(mh);
        mxFree(ih);
    } else if (is("estimate_pose_dlt")) {
        // M_tilde = vohip_mex('estimate_pose_dlt', p n x 2, P n x 3, K)   (estimatePoseDLT.m:1)
        const int n = rows(in[0]);
        if (cols(in[0]) != 2 || cols(in[1]) != 3 || rows(in[1]) != n) mexErrMsgIdAndTxt("vohip:size", "p must be n x 2 and P n x 3");
        plhs[0] = mxCreateDoubleMatrix(3, 4, mxREAL);
        check(vohip_estimate_pose_dlt(c, dbl(in[0], "p"), dbl(in[1], "P"), n, dbl(in[2], "K"), mxGetDoubles(plhs[0])));
    } else if (is("twist2homog")) {
        plhs[0] = mxCreateDoubleMatrix(4, 4, mxREAL);
        check(vohip_twist2homog(c, dbl(in[0], "twist"), 1, mxGetDoubles(plhs[0])));
    } else if (is("homog2twist")) {
        plhs[0] = mxCreateDoubleMatrix(6, 1, mxREAL);
        check(vohip_homog2twist(c, dbl(in[0], "H"), 1, mxGetDoubles(plhs[0])));
    } else if (is("p3p")) {
        // poses = vohip_mex('p3p', worldPoints 3x3, imageVectors 3x3): 3x16 [C1 R1 C2 R2 C3 R3 C4 R4] (p3p.m:41)
        if (numel(in[0]) != 9 || numel(in[1]) != 9) mexErrMsgIdAndTxt("vohip:size", "p3p takes two 3x3 matrices");
        plhs[0] = mxCreateDoubleMatrix(3, 16, mxREAL);
        check(vohip_p3p(c, dbl(in[0], "worldPoints"), dbl(in[1], "imageVectors"), 1, mxGetDoubles(plhs[0])));
    } else if (is("triangulate")) {
        // P = vohip_mex('triangulate', p1 3xn, p2 3xn, M1 3x4 or 3x4xn, M2 3x4 or 3x4xn)
        const int n = cols(in[0]);
        plhs[0] = mxCreateDoubleMatrix(4, n, mxREAL);
        check(vohip_triangulate(c, dbl(in[0], "p1"), dbl(in[1], "p2"), dbl(in[2], "M1"), dbl(in[3], "M2"),
                                numel(in[2]) > 12, numel(in[3]) > 12, n, mxGetDoubles(plhs[0])));
    } else if (is("triangulate_refine")) {
        const int n = cols(in[0]);
        plhs[0] = mxCreateDoubleMatrix(4, n, mxREAL);
        check(vohip_triangulate_refine(c, dbl(in[0], "P_init"), dbl(in[1], "p1"), dbl(in[2], "p2"), rows(in[1]),
                                       dbl(in[3], "M1"), dbl(in[4], "M2"), n, mxGetDoubles(plhs[0])));
    } else if (is("candidate_filter")) {
        // keep = vohip_mex('candidate_filter', Location (n x 2 single), S.C, S.P, r)
        if (!mxIsSingle(in[0])) mexErrMsgIdAndTxt("vohip:type", "Location must be single");
        const int n = rows(in[0]);
        mxArray *keep = mxCreateLogicalMatrix(n, 1);
        check(vohip_candidate_filter(c, mxGetSingles(in[0]), n, dbl(in[1], "S.C"), cols(in[1]), dbl(in[2], "S.P"),
                                     cols(in[2]), mxGetScalar(in[3]), (uint8_t *)mxGetLogicals(keep)));
        plhs[0] = keep;
    } else if (is("candidate_triangulate")) {
        // [status, X, angles] = vohip_mex('candidate_triangulate', cur_rc, first_rc, S.T, T_W_C, K, angle, min_d, max_d, is_BA)
        const int n = cols(in[0]);
        uint8_t *st = scratch<uint8_t>((size_t)n);
        plhs[1] = mxCreateDoubleMatrix(3, n, mxREAL);
        plhs[2] = mxCreateDoubleMatrix(1, n, mxREAL);
        const int rc = vohip_candidate_triangulate(c, dbl(in[0], "cur"), dbl(in[1], "first"), dbl(in[2], "S.T"), n,
                                                   dbl(in[3], "T_W_C"), dbl(in[4], "K"), mxGetScalar(in[5]),
                                                   mxGetScalar(in[6]), mxGetScalar(in[7]), mxGetScalar(in[8]) != 0,
                                                   st, mxGetDoubles(plhs[1]), mxGetDoubles(plhs[2]));
        check(rc);
        plhs[0] = mxCreateDoubleMatrix(1, n, mxREAL);
        for (int i = 0; i < n; ++i) mxGetDoubles(plhs[0])[i] = st[i];
        mxFree(st);
    } else if (is("fmatrix_ransac")) {
        // [F, inliers] = vohip_mex('fmatrix_ransac', p1 (n x 2), p2 (n x 2), samples 8 x T (1-based), threshold, confidence)
        // MATLAB's N-by-2 matrices are [x...; y...] in memory; the library wants 2 x n, so transpose here.
        const int n = rows(in[0]), T = cols(in[2]);
        double *a = interleave2(dbl(in[0], "p1"), n), *b = interleave2(dbl(in[1], "p2"), n);
        int32_t *s = idx0(in[2]);
        mxArray *inl = mxCreateLogicalMatrix(n, 1);
        double F[9];
        int best = 0, trials = 0;
        const int rc = vohip_fmatrix_ransac(c, a, b, n, s, T, mxGetScalar(in[3]), mxGetScalar(in[4]),
                                            (uint8_t *)mxGetLogicals(inl), &best, &trials, F);
        check(rc, true);
        mxFree(a);
        mxFree(b);
        mxFree(s);
        plhs[0] = mxCreateDoubleMatrix(3, 3, mxREAL);
        for (int r = 0; r < 3; ++r) for (int q = 0; q < 3; ++q) mxGetDoubles(plhs[0])[3 * q + r] = F[3 * r + q];
        plhs[1] = inl;
    } else if (is("align_sim3")) {
        // [p_G_C, x] = vohip_mex('align_sim3', pp_G_C 3xN, p_V_C 3xN)
        const int n = cols(in[0]);
        plhs[0] = mxCreateDoubleMatrix(3, n, mxREAL);
        plhs[1] = mxCreateDoubleMatrix(7, 1, mxREAL);
        check(vohip_align_sim3(c, dbl(in[0], "pp_G_C"), dbl(in[1], "p_V_C"), n, 400, 1e-6, 1e-6,   // lsqnonlin defaults (alignEstimateToGroundTruth.m:17-18)
                               mxGetDoubles(plhs[1]), mxGetDoubles(plhs[0]), nullptr));
    } else if (is("relative_pose")) {
        // [orientation, location, validFraction] = vohip_mex('relative_pose', E, K, p1 (n x 2), p2 (n x 2))
        const int n = rows(in[2]);
        double *a = interleave2(dbl(in[2], "p1"), n), *b = interleave2(dbl(in[3], "p2"), n);
        plhs[0] = mxCreateDoubleMatrix(3, 3, mxREAL);
        plhs[1] = mxCreateDoubleMatrix(1, 3, mxREAL);
        double frac = 0.0;
        const int rc = vohip_relative_pose(c, dbl(in[0], "E"), dbl(in[1], "K"), a, b, n, mxGetDoubles(plhs[0]),
                                           mxGetDoubles(plhs[1]), &frac);
        check(rc);
        mxFree(a);
        mxFree(b);
        plhs[2] = mxCreateDoubleScalar(frac);
    } else if (is("ba_error")) {
        int nobs = 0;
        double *r = scratch<double>((size_t)2 * numel(in[1]));
        check(vohip_ba_error(c, dbl(in[0], "state"), numel(in[0]), dbl(in[1], "observations"), numel(in[1]),
                             dbl(in[2], "K"), r, &nobs));
        plhs[0] = mxCreateDoubleMatrix(2, nobs, mxREAL);
        std::memcpy(mxGetDoubles(plhs[0]), r, sizeof(double) * 2 * nobs);
        mxFree(r);
    } else if (is("ba_run")) {
        // state = vohip_mex('ba_run', state, observations, K, max_iter)
        plhs[0] = mxDuplicateArray(in[0]);
        check(vohip_ba_run(c, mxGetDoubles(plhs[0]), numel(in[0]), dbl(in[1], "observations"), numel(in[1]),
                           dbl(in[2], "K"), (int)mxGetScalar(in[3]), 1e-6, 1e-6, nullptr));
    } else if (is("pose_refine")) {
        // T = vohip_mex('pose_refine', T_W_C, keypoints_uv 2xN, landmarks 3xN, K, max_iter)
        plhs[0] = mxDuplicateArray(in[0]);
        check(vohip_pose_refine(c, mxGetDoubles(plhs[0]), dbl(in[1], "keypoints"), dbl(in[2], "landmarks"), cols(in[1]),
                                dbl(in[3], "K"), (int)mxGetScalar(in[4]), 1e-6, 1e-6, nullptr));   // lsqnonlin's defaults (T_refinement.m:4)
    } else if (is("detect_corners")) {
        // [Location, Metric] = vohip_mex('detect_corners', image uint8, min_quality, filter_size, border, method)
        if (!mxIsUint8(in[0])) mexErrMsgIdAndTxt("vohip:type", "image must be uint8");
        const int cap = 16384;
        float *xy = scratch<float>((size_t)2 * cap), *metric = scratch<float>((size_t)cap);
        int n = 0;
        check(vohip_detect_corners(c, mxGetUint8s(in[0]), rows(in[0]), cols(in[0]), 1, mxGetScalar(in[1]),
                                   (int)mxGetScalar(in[2]), (int)mxGetScalar(in[3]), (int)mxGetScalar(in[4]), xy, metric,
                                   cap, &n));
        plhs[0] = mxCreateNumericMatrix(n, 2, mxSINGLE_CLASS, mxREAL);
        plhs[1] = mxCreateNumericMatrix(n, 1, mxSINGLE_CLASS, mxREAL);
        std::memcpy(mxGetSingles(plhs[0]), xy, sizeof(float) * 2 * n);   // n x 2, x column then y column
        std::memcpy(mxGetSingles(plhs[1]), metric, sizeof(float) * n);
        mxFree(xy);
        mxFree(metric);
    } else if (is("detect_fast")) {
        // [Location, Metric] = vohip_mex('detect_fast', image uint8, min_contrast, min_quality, border)
        if (!mxIsUint8(in[0])) mexErrMsgIdAndTxt("vohip:type", "image must be uint8");
        const int cap = 16384;
        float *xy = scratch<float>((size_t)2 * cap), *metric = scratch<float>((size_t)cap);
        int n = 0;
        check(vohip_detect_fast(c, mxGetUint8s(in[0]), rows(in[0]), cols(in[0]), 1, mxGetScalar(in[1]), mxGetScalar(in[2]),
                                (int)mxGetScalar(in[3]), xy, metric, cap, &n));
        plhs[0] = mxCreateNumericMatrix(n, 2, mxSINGLE_CLASS, mxREAL);
        plhs[1] = mxCreateNumericMatrix(n, 1, mxSINGLE_CLASS, mxREAL);
        std::memcpy(mxGetSingles(plhs[0]), xy, sizeof(float) * 2 * n);   // n x 2, x column then y column
        std::memcpy(mxGetSingles(plhs[1]), metric, sizeof(float) * n);
        mxFree(xy);
        mxFree(metric);
    } else if (is("detect_surf")) {
        // [Location, Metric, Scale] = vohip_mex('detect_surf', image uint8, metric_threshold, border)
        if (!mxIsUint8(in[0])) mexErrMsgIdAndTxt("vohip:type", "image must be uint8");
        const int cap = 16384;
        float *xy = scratch<float>((size_t)2 * cap), *metric = scratch<float>((size_t)cap), *scale = scratch<float>((size_t)cap);
        int n = 0;
        check(vohip_detect_surf(c, mxGetUint8s(in[0]), rows(in[0]), cols(in[0]), 1, mxGetScalar(in[1]), (int)mxGetScalar(in[2]),
                                xy, metric, scale, cap, &n));
        plhs[0] = mxCreateNumericMatrix(n, 2, mxSINGLE_CLASS, mxREAL);
        plhs[1] = mxCreateNumericMatrix(n, 1, mxSINGLE_CLASS, mxREAL);
        std::memcpy(mxGetSingles(plhs[0]), xy, sizeof(float) * 2 * n);   // n x 2, x column then y column
        std::memcpy(mxGetSingles(plhs[1]), metric, sizeof(float) * n);
        if (nlhs > 2) {
            plhs[2] = mxCreateNumericMatrix(n, 1, mxSINGLE_CLASS, mxREAL);
            std::memcpy(mxGetSingles(plhs[2]), scale, sizeof(float) * n);
        }
        mxFree(xy);
        mxFree(metric);
        mxFree(scale);
    } else if (is("detect_brisk")) {
        // [Location, Metric, Scale] = vohip_mex('detect_brisk', image uint8, min_contrast, min_quality, num_octaves, border)
        if (!mxIsUint8(in[0])) mexErrMsgIdAndTxt("vohip:type", "image must be uint8");
        const int cap = 16384;
        float *xy = scratch<float>((size_t)2 * cap), *metric = scratch<float>((size_t)cap), *scale = scratch<float>((size_t)cap);
        int n = 0;
        check(vohip_detect_brisk(c, mxGetUint8s(in[0]), rows(in[0]), cols(in[0]), 1, mxGetScalar(in[1]), mxGetScalar(in[2]),
                                 (int)mxGetScalar(in[3]), (int)mxGetScalar(in[4]), xy, metric, scale, cap, &n));
        plhs[0] = mxCreateNumericMatrix(n, 2, mxSINGLE_CLASS, mxREAL);
        plhs[1] = mxCreateNumericMatrix(n, 1, mxSINGLE_CLASS, mxREAL);
        std::memcpy(mxGetSingles(plhs[0]), xy, sizeof(float) * 2 * n);   // n x 2, x column then y column
        std::memcpy(mxGetSingles(plhs[1]), metric, sizeof(float) * n);
        if (nlhs > 2) {
            plhs[2] = mxCreateNumericMatrix(n, 1, mxSINGLE_CLASS, mxREAL);
            std::memcpy(mxGetSingles(plhs[2]), scale, sizeof(float) * n);
        }
        mxFree(xy);
        mxFree(metric);
        mxFree(scale);
    } else if (is("surf_features")) {
        // [features (N x 64 single), orientation] = vohip_mex('surf_features', image uint8, Location (N x 2 single), Scale, upright)
        if (!mxIsUint8(in[0]) || !mxIsSingle(in[1]) || !mxIsSingle(in[2]))
            mexErrMsgIdAndTxt("vohip:type", "image must be uint8, Location and Scale single");
        const int n = rows(in[1]);
        plhs[0] = mxCreateNumericMatrix(n, 64, mxSINGLE_CLASS, mxREAL);
        mxArray *ori = mxCreateNumericMatrix(n, 1, mxSINGLE_CLASS, mxREAL);
        if (n > 0) {
            float *f = scratch<float>((size_t)64 * n);   // a point's 64 numbers contiguous: the transpose of MATLAB's array
            check(vohip_extract_surf_features(c, mxGetUint8s(in[0]), rows(in[0]), cols(in[0]), 1, mxGetSingles(in[1]),
                                              mxGetSingles(in[2]), n, nrhs > 4 ? (int)mxGetScalar(in[3]) : 0, f,
                                              mxGetSingles(ori)));
            float *out = mxGetSingles(plhs[0]);
            for (int k = 0; k < 64; ++k)
                for (int i = 0; i < n; ++i) out[(size_t)k * n + i] = f[(size_t)i * 64 + k];
            mxFree(f);
        }
        if (nlhs > 1) plhs[1] = ori;
    } else if (is("match_features")) {
        // [indexPairs (P x 2 double, 1-based), metric] = vohip_mex('match_features', features1, features2, MatchThreshold, MaxRatio)
        if (!mxIsSingle(in[0]) || !mxIsSingle(in[1]) || cols(in[0]) != cols(in[1]))
            mexErrMsgIdAndTxt("vohip:type", "features must be single N1 x D and N2 x D");
        const int n0 = rows(in[0]), n1 = rows(in[1]), dim = cols(in[0]);
        int m = 0;
        int32_t *pairs = scratch<int32_t>((size_t)2 * (n0 + 1));
        float *metric = scratch<float>((size_t)n0 + 1);
        if (n0 > 0 && n1 > 0) {
            float *f0 = scratch<float>((size_t)n0 * dim), *f1 = scratch<float>((size_t)n1 * dim);
            const float *a = mxGetSingles(in[0]), *b = mxGetSingles(in[1]);
            for (int i = 0; i < n0; ++i)
                for (int k = 0; k < dim; ++k) f0[(size_t)i * dim + k] = a[(size_t)k * n0 + i];
            for (int i = 0; i < n1; ++i)
                for (int k = 0; k < dim; ++k) f1[(size_t)i * dim + k] = b[(size_t)k * n1 + i];
            check(vohip_match_features(c, f0, n0, f1, n1, dim, nrhs > 3 ? mxGetScalar(in[2]) : 1.0,
                                       nrhs > 4 ? mxGetScalar(in[3]) : 0.6, pairs, metric, &m));
            mxFree(f0);
            mxFree(f1);
        }
        plhs[0] = mxCreateDoubleMatrix(m, 2, mxREAL);
        double *ip = mxGetDoubles(plhs[0]);
        for (int i = 0; i < m; ++i) {
            ip[i] = pairs[2 * i] + 1.0;
            ip[m + i] = pairs[2 * i + 1] + 1.0;
        }
        if (nlhs > 1) {
            plhs[1] = mxCreateNumericMatrix(m, 1, mxSINGLE_CLASS, mxREAL);
            std::memcpy(mxGetSingles(plhs[1]), metric, sizeof(float) * m);
        }
        mxFree(pairs);
        mxFree(metric);
    } else if (is("pyramid_build")) {
        // vohip_mex('pyramid_build', slot, image uint8, levels)
        if (!mxIsUint8(in[1])) mexErrMsgIdAndTxt("vohip:type", "image must be uint8");
        check(vohip_pyramid_build(c, (int)mxGetScalar(in[0]), mxGetUint8s(in[1]), rows(in[1]), cols(in[1]), 1,
                                  (int)mxGetScalar(in[2])));
    } else if (is("klt_track")) {
        // [points, validity] = vohip_mex('klt_track', slot_prev, slot_cur, points (N x 2), block, max_iter, max_bidir)
        const int n = rows(in[2]);
        plhs[0] = mxCreateDoubleMatrix(n, 2, mxREAL);
        mxArray *valid = mxCreateLogicalMatrix(n, 1);
        check(vohip_klt_track(c, (int)mxGetScalar(in[0]), (int)mxGetScalar(in[1]), dbl(in[2], "points"), n,
                              (int)mxGetScalar(in[3]), (int)mxGetScalar(in[4]), mxGetScalar(in[5]),
                              mxGetDoubles(plhs[0]), (uint8_t *)mxGetLogicals(valid), nullptr));
        plhs[1] = valid;
    } else if (is("frame_create")) {
        // vohip_mex('frame_create', K, rows, cols, params struct): the loop body of main_demo.m:377-427 as
        // one call per frame, state resident on the GPU (one loop per MATLAB session).  params fields
        // (all optional): r_discard_redundant, angle_threshold_deg, min_depth, max_depth, min_quality,
        // filter_size, detector, min_contrast, metric_threshold, brisk_octaves, refine_pose, is_BA.
        if (g_frame) { vohip_frame_destroy(g_frame); g_frame = nullptr; }
        vohip_frame_params p;
        vohip_frame_default_params(&p);
        if (nrhs > 4 && mxIsStruct(in[3])) frame_params_from_struct(in[3], p);
        check(vohip_frame_create(c, dbl(in[0], "K"), (int)mxGetScalar(in[1]), (int)mxGetScalar(in[2]), 1, &p, &g_frame));
    } else if (is("frame_reset")) {
        // vohip_mex('frame_reset', image uint8, X 3xn, P_rc 2xn, T_W_C 4x4)   (main_demo.m:268-303, 342-344)
        if (!g_frame) mexErrMsgIdAndTxt("vohip:usage", "frame_create first");
        if (!mxIsUint8(in[0])) mexErrMsgIdAndTxt("vohip:type", "image must be uint8");
        check(vohip_frame_reset(g_frame, mxGetUint8s(in[0]), 0, dbl(in[1], "X"), nullptr, dbl(in[2], "P"), cols(in[1]),
                                dbl(in[3], "T_W_C")));
    } else if (is("frame_step")) {
        // [T_W_C, info, status] = vohip_mex('frame_step', image uint8, u3 3xh, u8 8x500 or [], next_image or [])
        // u3 / u8: uniform draws (rand); status 0 = done, VOHIP_SAMPLES_EXHAUSTED = call 'frame_retry' with a
        // longer u3 that starts with the same columns (info.num_iterations says how many are wanted).
        if (!g_frame) mexErrMsgIdAndTxt("vohip:usage", "frame_create first");
        if (!mxIsUint8(in[0])) mexErrMsgIdAndTxt("vohip:type", "image must be uint8");
        const bool have_u8 = nrhs > 3 && !mxIsEmpty(in[2]);
        const bool have_next = nrhs > 4 && !mxIsEmpty(in[3]);
        if (have_next && !mxIsUint8(in[3])) mexErrMsgIdAndTxt("vohip:type", "next image must be uint8");
        vohip_frame_pose pose;
        vohip_frame_counts counts;
        const int st = vohip_frame_step_ahead(g_frame, mxGetUint8s(in[0]), 0, have_next ? mxGetUint8s(in[3]) : nullptr, 0,
                                              dbl(in[1], "u3"), cols(in[1]), have_u8 ? dbl(in[2], "u8") : nullptr, &pose,
                                              &counts);
        if (st != VOHIP_OK && st != VOHIP_SAMPLES_EXHAUSTED) check(st);
        frame_outputs(nlhs, plhs, pose, counts, st);
    } else if (is("frame_retry")) {
        // [T_W_C, info, status] = vohip_mex('frame_retry', u3 3xh)
        if (!g_frame) mexErrMsgIdAndTxt("vohip:usage", "frame_create first");
        vohip_frame_pose pose;
        vohip_frame_counts counts;
        const int st = vohip_frame_retry(g_frame, dbl(in[0], "u3"), cols(in[0]), &pose, &counts);
        if (st != VOHIP_OK && st != VOHIP_SAMPLES_EXHAUSTED) check(st);
        frame_outputs(nlhs, plhs, pose, counts, st);
    } else if (is("frame_state")) {
        // [P, X, C, F, T] = vohip_mex('frame_state'): S.P 2xn, S.X 4xn (ids in row 4), S.C, S.F 2xm, S.T 12xm
        if (!g_frame) mexErrMsgIdAndTxt("vohip:usage", "frame_create first");
        int nL = 0, nC = 0;
        check(vohip_frame_sizes(g_frame, &nL, &nC));
        plhs[0] = mxCreateDoubleMatrix(2, nL, mxREAL);
        mxArray *X3 = mxCreateDoubleMatrix(3, nL, mxREAL), *ids = mxCreateDoubleMatrix(1, nL, mxREAL);
        mxArray *C = mxCreateDoubleMatrix(2, nC, mxREAL), *Fm = mxCreateDoubleMatrix(2, nC, mxREAL);
        mxArray *T = mxCreateDoubleMatrix(12, nC, mxREAL);
        check(vohip_frame_state(g_frame, mxGetDoubles(plhs[0]), mxGetDoubles(X3), mxGetDoubles(ids), mxGetDoubles(C),
                                mxGetDoubles(Fm), mxGetDoubles(T)));
        plhs[1] = mxCreateDoubleMatrix(4, nL, mxREAL);
        for (int k = 0; k < nL; ++k) {
            std::memcpy(mxGetDoubles(plhs[1]) + 4 * (size_t)k, mxGetDoubles(X3) + 3 * (size_t)k, sizeof(double) * 3);
            mxGetDoubles(plhs[1])[4 * (size_t)k + 3] = mxGetDoubles(ids)[k];
        }
        mxDestroyArray(X3);
        mxDestroyArray(ids);
        plhs[2] = C;
        plhs[3] = Fm;
        plhs[4] = T;
    } else if (is("frame_apply_ba")) {
        // vohip_mex('frame_apply_ba', X 3xn, T_W_C 4x4): VO_bundle_adjust.m:53-70 put back into the device state
        if (!g_frame) mexErrMsgIdAndTxt("vohip:usage", "frame_create first");
        check(vohip_frame_apply_ba(g_frame, dbl(in[0], "X"), cols(in[0]), dbl(in[1], "T_W_C")));
    } else if (is("window_create")) {
        // vohip_mex('window_create', K, params struct): the struct B of main_demo.m:288-303 kept by the library.
        // params fields (all optional): window_size, keyframe_d, min_depth, max_depth.
        if (g_window) { vohip_window_destroy(g_window); g_window = nullptr; }
        vohip_window_params p;
        vohip_window_default_params(&p);
        if (nrhs > 2 && mxIsStruct(in[1])) {
            auto field = [&](const char *name, double dflt) {
                const mxArray *f = mxGetField(in[1], 0, name);
                return f ? mxGetScalar(f) : dflt;
            };
            p.window_size = (int)field("window_size", p.window_size);
            p.keyframe_d = (int)field("keyframe_d", p.keyframe_d);
            p.min_depth = field("min_depth", p.min_depth);
            p.max_depth = field("max_depth", p.max_depth);
        }
        check(vohip_window_create(c, dbl(in[0], "K"), &p, &g_window));
    } else if (is("window_reset")) {
        // vohip_mex('window_reset', X 3xn, P_rc 2xn, T_W_C 4x4)   (main_demo.m:288-303)
        if (!g_window) mexErrMsgIdAndTxt("vohip:usage", "window_create first");
        check(vohip_window_reset(g_window, dbl(in[0], "X"), dbl(in[1], "P"), cols(in[0]), dbl(in[2], "T_W_C")));
    } else if (is("window_step") || is("frame_bundle_adjust")) {
        // [X 3xn, optimised, offsets 1xk, poses 16xk, num_key] = vohip_mex('window_step', S.P 2xn, S.X 4xn, T_W_C 4x4)
        // [optimised, offsets 1xk, poses 16xk, num_key] = vohip_mex('frame_bundle_adjust', T_W_C 4x4)
        // VO_bundle_adjust.m:1-98; offsets count columns of S.est_trans / S.est_rot back from the last one,
        // poses holds the 4x4 T_W_C (column-major) that belongs in each of them.
        if (!g_window) mexErrMsgIdAndTxt("vohip:usage", "window_create first");
        const int cap = 64;   // window_size <= 8: at most 8 + 7 keyframe_d poses
        int *offsets = scratch<int>(cap);
        double *poses = scratch<double>((size_t)16 * cap);
        int optimised = 0, n_poses = 0, out = 0;
        if (is("window_step")) {
            const int n = cols(in[1]);
            const double *SX = dbl(in[1], "S.X");
            if (rows(in[1]) != 4 || cols(in[0]) != n) mexErrMsgIdAndTxt("vohip:usage", "S.P 2xn and S.X 4xn are needed");
            double *ids = scratch<double>((size_t)n);
            plhs[0] = mxCreateDoubleMatrix(3, n, mxREAL);
            double *X = mxGetDoubles(plhs[0]);
            for (int k = 0; k < n; ++k) {
                std::memcpy(X + 3 * (size_t)k, SX + 4 * (size_t)k, sizeof(double) * 3);
                ids[k] = SX[4 * (size_t)k + 3];
            }
            check(vohip_window_step(g_window, dbl(in[0], "S.P"), X, ids, n, dbl(in[2], "T_W_C"), &optimised, &n_poses,
                                    offsets, poses, cap));
            out = 1;
        } else {
            if (!g_frame) mexErrMsgIdAndTxt("vohip:usage", "frame_create first");
            check(vohip_frame_bundle_adjust(g_frame, g_window, dbl(in[0], "T_W_C"), &optimised, &n_poses, offsets, poses, cap));
        }
        if (nlhs > out) plhs[out] = mxCreateDoubleScalar(optimised);
        if (nlhs > out + 1) {
            plhs[out + 1] = mxCreateDoubleMatrix(1, n_poses, mxREAL);
            for (int k = 0; k < n_poses; ++k) mxGetDoubles(plhs[out + 1])[k] = offsets[k];
        }
        if (nlhs > out + 2) {
            plhs[out + 2] = mxCreateDoubleMatrix(16, n_poses, mxREAL);
            std::memcpy(mxGetDoubles(plhs[out + 2]), poses, sizeof(double) * 16 * (size_t)n_poses);
        }
        if (nlhs > out + 3) {
            int num_key = 0;
            check(vohip_window_state(g_window, &num_key, nullptr, nullptr, nullptr));
            plhs[out + 3] = mxCreateDoubleScalar(num_key);
        }
    } else if (is("group_create")) {
        // vohip_mex('group_create', K, rows, cols, params struct, n_sequences, max_frames): n_sequences loops of one
        // image size, camera and parameter set that advance together, one launch sequence per frame for all of them
        // (the independent runs of main_eval.m:20-23); params as for 'frame_create' (is_BA must be 0).
        if (g_group) { vohip_group_destroy(g_group); g_group = nullptr; }
        vohip_frame_params p;
        vohip_frame_default_params(&p);
        if (mxIsStruct(in[3])) frame_params_from_struct(in[3], p);
        g_group_n = (int)mxGetScalar(in[4]);
        check(vohip_group_create(c, dbl(in[0], "K"), (int)mxGetScalar(in[1]), (int)mxGetScalar(in[2]), 1, &p, g_group_n,
                                 (int)mxGetScalar(in[5]), &g_group));
    } else if (is("group_prepare")) {
        // vohip_mex('group_prepare', seq (1-based), images (cell array of uint8; the first is the image the loop is
        // reset on, main_demo.m:342-344), detector (-1: the group's, 0 Harris, 1 MinEigen, 2 FAST, 3 SURF, 4 BRISK), min_quality, min_contrast (SURF: metric_threshold))
        // vohip_mex('group_prepare', seq, from_seq (a sequence that holds the same images), detector, min_quality, min_contrast)
        if (!g_group) mexErrMsgIdAndTxt("vohip:usage", "group_create first");
        const int seq = (int)mxGetScalar(in[0]) - 1;
        if (mxIsCell(in[1])) {
            const int n = numel(in[1]);
            const uint8_t **imgs = scratch<const uint8_t *>((size_t)n);
            for (int k = 0; k < n; ++k) {
                const mxArray *im = mxGetCell(in[1], k);
                if (!im || !mxIsUint8(im)) mexErrMsgIdAndTxt("vohip:type", "images must be uint8");
                imgs[k] = mxGetUint8s(im);
            }
            check(vohip_group_prepare(g_group, seq, imgs, n, 0, (int)mxGetScalar(in[2]), mxGetScalar(in[3]), mxGetScalar(in[4])));
        } else {
            check(vohip_group_prepare_like(g_group, seq, (int)mxGetScalar(in[1]) - 1, (int)mxGetScalar(in[2]), mxGetScalar(in[3]),
                                           mxGetScalar(in[4])));
        }
    } else if (is("group_reset")) {
        // vohip_mex('group_reset', seq, X 3xn, P_rc 2xn, T_W_C 4x4)   (the state after the bootstrap, main_demo.m:268-303)
        if (!g_group) mexErrMsgIdAndTxt("vohip:usage", "group_create first");
        check(vohip_group_reset(g_group, (int)mxGetScalar(in[0]) - 1, dbl(in[1], "X"), nullptr, dbl(in[2], "P"), cols(in[1]),
                                dbl(in[3], "T_W_C")));
    } else if (is("group_step")) {
        // [T_W_C 4x4xS, info 1xS, status 1xS] = vohip_mex('group_step', u3 3 x h x S, u8 8 x trials x S or [])
        // One frame of every sequence that has one left; status(k): 0 done, VOHIP_SAMPLES_EXHAUSTED -> 'group_retry'
        // for that sequence, VOHIP_SEQUENCE_IDLE the sequence took no part, else it stopped (as the reference's
        // script does at main_demo.m:389-391) and stays idle.
        if (!g_group) mexErrMsgIdAndTxt("vohip:usage", "group_create first");
        const int S = g_group_n;
        const mwSize *d3 = mxGetDimensions(in[0]);
        const int h = mxGetNumberOfDimensions(in[0]) >= 2 ? (int)d3[1] : 0;
        if (rows(in[0]) != 3 || numel(in[0]) != 3 * h * S) mexErrMsgIdAndTxt("vohip:usage", "u3 must be 3 x h x n_sequences");
        const bool have_u8 = nrhs > 2 && !mxIsEmpty(in[1]);
        check(vohip_group_begin(g_group, dbl(in[0], "u3"), h, have_u8 ? dbl(in[1], "u8") : nullptr));
        vohip_frame_pose *pose = scratch<vohip_frame_pose>((size_t)S);
        vohip_frame_counts *counts = scratch<vohip_frame_counts>((size_t)S);
        int *status = scratch<int>((size_t)S);
        check(vohip_group_finish(g_group, pose, counts, status));
        const mwSize dims[3] = {4, 4, (mwSize)S};
        plhs[0] = mxCreateNumericArray(3, dims, mxDOUBLE_CLASS, mxREAL);
        for (int s = 0; s < S; ++s) std::memcpy(mxGetDoubles(plhs[0]) + 16 * (size_t)s, pose[s].T_W_C, sizeof(double) * 16);
        if (nlhs > 1) plhs[1] = group_info(pose, counts, S);
        if (nlhs > 2) {
            plhs[2] = mxCreateDoubleMatrix(1, S, mxREAL);
            for (int s = 0; s < S; ++s) mxGetDoubles(plhs[2])[s] = status[s];
        }
    } else if (is("group_retry")) {
        // [T_W_C, info, status] = vohip_mex('group_retry', seq, u3 3xh): more hypotheses for one sequence
        if (!g_group) mexErrMsgIdAndTxt("vohip:usage", "group_create first");
        vohip_frame_pose pose;
        vohip_frame_counts counts;
        const int st = vohip_group_retry(g_group, (int)mxGetScalar(in[0]) - 1, dbl(in[1], "u3"), cols(in[1]), &pose, &counts);
        if (st != VOHIP_OK && st != VOHIP_SAMPLES_EXHAUSTED) check(st);
        frame_outputs(nlhs, plhs, pose, counts, st);
    } else if (is("group_state")) {
        // [P, X, C, F, T] = vohip_mex('group_state', seq): as 'frame_state', of one sequence
        if (!g_group) mexErrMsgIdAndTxt("vohip:usage", "group_create first");
        const int seq = (int)mxGetScalar(in[0]) - 1;
        int nL = 0, nC = 0;
        check(vohip_group_sizes(g_group, seq, &nL, &nC, nullptr));
        plhs[0] = mxCreateDoubleMatrix(2, nL, mxREAL);
        mxArray *X3 = mxCreateDoubleMatrix(3, nL, mxREAL), *ids = mxCreateDoubleMatrix(1, nL, mxREAL);
        mxArray *C = mxCreateDoubleMatrix(2, nC, mxREAL), *Fm = mxCreateDoubleMatrix(2, nC, mxREAL);
        mxArray *T = mxCreateDoubleMatrix(12, nC, mxREAL);
        check(vohip_group_state(g_group, seq, mxGetDoubles(plhs[0]), mxGetDoubles(X3), mxGetDoubles(ids), mxGetDoubles(C),
                                mxGetDoubles(Fm), mxGetDoubles(T)));
        plhs[1] = mxCreateDoubleMatrix(4, nL, mxREAL);
        for (int k = 0; k < nL; ++k) {
            std::memcpy(mxGetDoubles(plhs[1]) + 4 * (size_t)k, mxGetDoubles(X3) + 3 * (size_t)k, sizeof(double) * 3);
            mxGetDoubles(plhs[1])[4 * (size_t)k + 3] = mxGetDoubles(ids)[k];
        }
        mxDestroyArray(X3);
        mxDestroyArray(ids);
        plhs[2] = C;
        plhs[3] = Fm;
        plhs[4] = T;
    } else if (is("group_destroy")) {
        if (g_group) vohip_group_destroy(g_group);
        g_group = nullptr;
    } else if (is("window_destroy")) {
        if (g_window) vohip_window_destroy(g_window);
        g_window = nullptr;
    } else if (is("frame_destroy")) {
        if (g_frame) vohip_frame_destroy(g_frame);
        g_frame = nullptr;
    } else {
        mexErrMsgIdAndTxt("vohip:usage", "unknown opcode '%s'", opbuf);
    }
}
