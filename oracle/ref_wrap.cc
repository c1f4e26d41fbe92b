// oracle/ref_wrap.cc -- plain-C entry points into the REFERENCE's own CPU code, for the checker only.
//
// TEST INFRASTRUCTURE ONLY.  Compiled by `make -C oracle ref` together with the reference sources where they lie
// under $(REF) into oracle/_ref/ (git-ignored).  Nothing of the reference is restated here: this file only builds
// the OFFlags argument (a C++ class with std::string members, which ctypes cannot pass by value) and forwards.
#include "offlags.h"   // from $(REF)/include
#include "oct_bc.h"    // from $(REF)/include: template <class T> T oct_bc(T x, int nx, bool &bc)   (ref include/oct_bc.h:1-20)

// ref src/oct_binterp.cc:24 and :36 (the bilinear weights and their re-use; the device copies .cu:56-71 repeat them in float)
double oct_binterp_coefs(double x, double y, double x1, double x2, double y1, double y2, double f11, double f21, double f12, double f22,
                         double &p1, double &p2, double &p3, double &p4);
double oct_coef_binterp(double p1, double p2, double p3, double p4, double f11, double f21, double f12, double f22);

extern "C" double oct_ref_binterp_coefs(double x, double y, double x1, double x2, double y1, double y2, double f11, double f21, double f12,
                                        double f22, double *p)
{
    return oct_binterp_coefs(x, y, x1, x2, y1, y2, f11, f21, f12, f22, p[0], p[1], p[2], p[3]);
}
extern "C" double oct_ref_coef_binterp(const double *p, double f11, double f21, double f12, double f22)
{
    return oct_coef_binterp(p[0], p[1], p[2], p[3], f11, f21, f12, f22);
}
extern "C" float oct_ref_bc_float(float x, int nx, int *hit) { bool b; const float r = oct_bc<float>(x, nx, b); *hit = b ? 1 : 0; return r; }
extern "C" double oct_ref_bc_double(double x, int nx, int *hit) { bool b; const double r = oct_bc<double>(x, nx, b); *hit = b ? 1 : 0; return r; }
extern "C" int oct_ref_bc_int(int x, int nx, int *hit) { bool b; const int r = oct_bc<int>(x, nx, b); *hit = b ? 1 : 0; return r; }

void oct_patch_match_optical_flow(float *, float *, float *, float *, int, int, OFFlags);   // ref src/oct_patch_match_optical_flow.cc:56

extern "C" void oct_ref_patch_match(const float *img1, const float *img2, float *u_inout, float *v_inout, int nx, int ny,
                                    int rad, int srad)
{
    OFFlags args;
    args.rad = rad;       // the only two fields the function reads (:68-69)
    args.srad = srad;
    oct_patch_match_optical_flow(const_cast<float *>(img1), const_cast<float *>(img2), u_inout, v_inout, nx, ny, args);
}

void oct_bandminmax(int gb, float &maxch, float &minch);   // ref src/oct_normalize_geo.cc:9 (the ABI band range table; bands it does not know leave both untouched)

extern "C" void oct_ref_bandminmax(int gb, float *maxch, float *minch) { oct_bandminmax(gb, *maxch, *minch); }
