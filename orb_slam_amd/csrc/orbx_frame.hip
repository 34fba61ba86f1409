// orbx_frame.hip -- per-keypoint maps that the reference's Frame runs right after extraction.
//
//   k_undistort   Frame::UndistortKeyPoints / UndistortKeyPointsStereo   (src/Frame.cc:745-852)
//
// The reference hands the keypoint coordinates to cv::undistortPoints (or, when built with
// USE_FISHEYE_DISTORTION, cv::fisheye::undistortPoints) and copies the other keypoint fields.
// Both are evaluated here in double, operation by operation as the OpenCV 3.4.1 sources write
// them (no contraction), and rounded to float once at the end.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/orbx.h"
#include "orbx_kernels.hpp"

namespace orbx {

// thread = one keypoint slot of one frame; grid (ceil(cap / 256), frames)
__global__ __launch_bounds__(256) void k_undistort(const orbx_keypoint* __restrict__ in, const int32_t* __restrict__ n_per_frame,
                                                   int n_fixed, int cap, UndistortModel m, orbx_keypoint* __restrict__ out)
{
    const int f = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int n = n_per_frame ? min(n_per_frame[f], cap) : n_fixed;
    if (i >= n) return;
    orbx_keypoint kp = in[(size_t)f * cap + i];
    double x, y;
    if (!m.fisheye) {
        // cv::undistortPoints: 5 fixed-point iterations of the inverse distortion
        x = ((double)kp.x - m.cx) * m.ifx;
        y = ((double)kp.y - m.cy) * m.ify;
        if (m.ndist > 0) {
            const double x0 = x, y0 = y;
            const double* k = m.k;
#pragma unroll 1
            for (int j = 0; j < 5; j++) {
                const double r2 = x * x + y * y;
                const double icdist = (1 + ((k[7] * r2 + k[6]) * r2 + k[5]) * r2) / (1 + ((k[4] * r2 + k[1]) * r2 + k[0]) * r2);
                const double deltaX = 2 * k[2] * x * y + k[3] * (r2 + 2 * x * x) + k[8] * r2 + k[9] * r2 * r2;
                const double deltaY = k[2] * (r2 + 2 * y * y) + 2 * k[3] * x * y + k[10] * r2 + k[11] * r2 * r2;
                x = (x0 - deltaX) * icdist;
                y = (y0 - deltaY) * icdist;
            }
        }
        const double xx = m.RR[0] * x + m.RR[1] * y + m.RR[2];
        const double yy = m.RR[3] * x + m.RR[4] * y + m.RR[5];
        const double ww = 1. / (m.RR[6] * x + m.RR[7] * y + m.RR[8]);
        kp.x = (float)(xx * ww);
        kp.y = (float)(yy * ww);
    } else {
        // cv::fisheye::undistortPoints: 10 fixed-point iterations for theta
        const double pwx = ((double)kp.x - m.cx) / m.fx, pwy = ((double)kp.y - m.cy) / m.fy;
        double scale = 1.0;
        double theta_d = sqrt(pwx * pwx + pwy * pwy);
        const double half_pi = 3.1415926535897932384626433832795 / 2.;
        theta_d = fmin(fmax(-half_pi, theta_d), half_pi);
        if (theta_d > 1e-8) {
            double theta = theta_d;
            const double* k = m.k;
#pragma unroll 1
            for (int j = 0; j < 10; j++) {
                const double t2 = theta * theta, t4 = t2 * t2, t6 = t4 * t2, t8 = t6 * t2;
                theta = theta_d / (1 + k[0] * t2 + k[1] * t4 + k[2] * t6 + k[3] * t8);
            }
            scale = tan(theta) / theta_d;
        }
        x = pwx * scale;
        y = pwy * scale;
        const double px = m.RR[0] * x + m.RR[1] * y + m.RR[2];
        const double py = m.RR[3] * x + m.RR[4] * y + m.RR[5];
        const double pz = m.RR[6] * x + m.RR[7] * y + m.RR[8];
        kp.x = (float)(px / pz);
        kp.y = (float)(py / pz);
    }
    out[(size_t)f * cap + i] = kp;
}

void launch_undistort(const orbx_keypoint* d_in, const int32_t* d_n, int n_fixed, int cap, int nframes,
                      const UndistortModel& m, orbx_keypoint* d_out, hipStream_t s)
{
    if (cap <= 0 || nframes <= 0) return;
    dim3 grid((cap + 255) / 256, nframes);
    hipLaunchKernelGGL(k_undistort, grid, dim3(256), 0, s, d_in, d_n, n_fixed, cap, m, d_out);
}

} // namespace orbx
