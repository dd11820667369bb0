// C entry points around the reference's own cfar.cpp (TEST INFRASTRUCTURE).  REF_CFAR_CPP is the
// absolute path of bruce_slam/src/bruce_slam/cpp/cfar.cpp under /root/reference; it is #included
// from where it lies (never copied) and compiled against the stand-in headers in oracle/ref_shim/.
// The wrapper does what pybind11 does at the Python boundary (cfar.cpp:10, CFAR.py:123-133): cast-copy
// the numpy image into a MatrixXf, call the function, hand the uint8 mask (and the float threshold
// map of the *2 variants) back.
#include REF_CFAR_CPP

extern "C" int ref_cfar(const float *img, int rows, int cols, int alg /*0 CA 1 SOCA 2 GOCA 3 OS*/, int train_hs,
                        int guard_hs, int k, double tau, int want_threshold, uint8_t *mask_out, float *thr_out)
{
    MatrixXf m(rows, cols);
    for (int r = 0; r < rows; ++r)
        for (int c = 0; c < cols; ++c)
            m(r, c) = img[(size_t)r * cols + c];
    MatrixXb mask;
    MatrixXf thr;
    if (!want_threshold) {
        switch (alg) {
        case 0: mask = ca(m, train_hs, guard_hs, tau); break;
        case 1: mask = soca(m, train_hs, guard_hs, tau); break;
        case 2: mask = goca(m, train_hs, guard_hs, tau); break;
        case 3: mask = os(m, train_hs, guard_hs, k, tau); break;
        default: return -1;
        }
    } else {
        std::pair<MatrixXb, MatrixXf> p;
        switch (alg) {
        case 0: p = ca2(m, train_hs, guard_hs, tau); break;
        case 1: p = soca2(m, train_hs, guard_hs, tau); break;
        case 2: p = goca2(m, train_hs, guard_hs, tau); break;
        case 3: p = os2(m, train_hs, guard_hs, k, tau); break;
        default: return -1;
        }
        mask = p.first;
        thr = p.second;
    }
    for (int r = 0; r < rows; ++r)
        for (int c = 0; c < cols; ++c) {
            mask_out[(size_t)r * cols + c] = mask(r, c);
            if (want_threshold)
                thr_out[(size_t)r * cols + c] = thr(r, c);
        }
    return 0;
}
