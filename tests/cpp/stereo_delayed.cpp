// The C++ mirror of the reference's delayed stereo matching (orbx::DelayedStereo in include/orbx_host.hpp)
// driven as Tracking drives a Frame: PrepareStereoCandidates, an online call, single-point calls, the offline
// call.  Input and output are flat binary files written / checked by tests/test_gpu_match.py.
//   stereo_delayed <in.bin> <out.bin>
// in : int32 nl, nr, nrows, npoints; float mbf, mb, minx; kl[nl] (28 B), dl[nl*32], kr[nr], dr[nr*32],
//      has_mappoint[nl] (u8), prior_disp[nl] (f32), then npoints x { int32 iL, th; float minD, maxD; uint8 add_mappoint }
// out: int32 nm_online, nm_offline, npoints x uint8 result; then mvuRight[nl], mvDepth[nl], listed_dist[nl], mvStereoMatched[nl]
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "orbx_host.hpp"

template <class T> static void rd(FILE* f, T* p, size_t n) { if (n && std::fread(p, sizeof(T), n, f) != n) { std::fprintf(stderr, "short input\n"); std::exit(2); } }

int main(int argc, char** argv)
{
    if (argc != 3) return 2;
    FILE* f = std::fopen(argv[1], "rb");
    if (!f) return 2;
    int32_t hdr[4];
    float par[3];
    rd(f, hdr, 4);
    rd(f, par, 3);
    const int nl = hdr[0], nr = hdr[1], nrows = hdr[2], np = hdr[3];
    std::vector<orbx_keypoint> kl(nl), kr(nr);
    std::vector<uint8_t> dl((size_t)nl * 32), dr((size_t)nr * 32), mp(nl);
    std::vector<float> prior(nl);
    rd(f, kl.data(), nl); rd(f, dl.data(), dl.size()); rd(f, kr.data(), nr); rd(f, dr.data(), dr.size());
    rd(f, mp.data(), nl); rd(f, prior.data(), nl);
    try {
        orbx::ORBextractor ex(1000, 1.2f, 8, 20, 7);
        orbx::DelayedStereo st;
        st.PrepareStereoCandidates(nl);
        st.has_mappoint = mp;
        st.prior_disp = prior;
        const int nm_on = st.ComputeStereoMatches_Undistorted(ex, true, kl, dl.data(), kr, dr.data(), nrows, par[0], par[1], par[2]);
        std::vector<uint8_t> res(np);
        for (int k = 0; k < np; k++) {
            int32_t a[2]; float b[2]; uint8_t add;
            rd(f, a, 2); rd(f, b, 2); rd(f, &add, 1);
            if (add && a[0] >= 0 && a[0] < nl) st.has_mappoint[a[0]] = 1;
            res[k] = st.ComputeStereoMatch_OnePoint(ex, a[0], a[1], b[0], b[1], kl, dl.data(), kr, dr.data(), nrows, par[0], par[1]);
        }
        const int nm_off = st.ComputeStereoMatches_Undistorted(ex, false, kl, dl.data(), kr, dr.data(), nrows, par[0], par[1], par[2]);
        FILE* o = std::fopen(argv[2], "wb");
        int32_t nm[2] = {nm_on, nm_off};
        std::fwrite(nm, 4, 2, o);
        std::fwrite(res.data(), 1, res.size(), o);
        std::fwrite(st.mvuRight.data(), 4, nl, o);
        std::fwrite(st.mvDepth.data(), 4, nl, o);
        std::fwrite(st.listed_dist.data(), 4, nl, o);
        std::fwrite(st.mvStereoMatched.data(), 1, nl, o);
        std::fclose(o);
    } catch (const std::exception& e) {
        std::fprintf(stderr, "%s\n", e.what());
        return 1;
    }
    std::fclose(f);
    return 0;
}
