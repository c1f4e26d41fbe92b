// oracle/ref_wrap.cc -- plain-C entry points into the REFERENCE's own CPU code, for the checker only.
//
// TEST INFRASTRUCTURE ONLY.  Compiled by `make -C oracle ref` together with the reference sources where they lie
// under $(REF) into oracle/_ref/ (git-ignored).  Nothing of the reference is restated here: this file only builds
// the OFFlags argument (a C++ class with std::string members, which ctypes cannot pass by value) and forwards.
#include "offlags.h"   // from $(REF)/include

void oct_patch_match_optical_flow(float *, float *, float *, float *, int, int, OFFlags);   // ref src/oct_patch_match_optical_flow.cc:56

extern "C" void oct_ref_patch_match(const float *img1, const float *img2, float *u_inout, float *v_inout, int nx, int ny,
                                    int rad, int srad)
{
    OFFlags args;
    args.rad = rad;       // the only two fields the function reads (:68-69)
    args.srad = srad;
    oct_patch_match_optical_flow(const_cast<float *>(img1), const_cast<float *>(img2), u_inout, v_inout, nx, ny, args);
}
