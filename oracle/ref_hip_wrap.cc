/* ref_hip_wrap.cc -- forwarding wrapper for the CROSS-CHECK build of the reference's two CUDA translation units (oracle/Makefile,
 * target `refhip`): src/oct_variational_optical_flow.cu and src/oct_pix2uv_cuda.cu are passed through the image's own hipify-perl WHERE
 * THEY LIE, the translated text goes to a scratch directory outside the repository (never committed, never shipped), hipcc builds it for
 * gfx950 together with this file into oracle/_ref/liboct_ref_hip.so.  TEST INFRASTRUCTURE ONLY, and a TOOL STAND-IN (hipify + hipcc in place of nvcc, ocml in
 * place of libdevice, 64-wide wavefronts in place of 32-wide warps): by the rules of this build it pins nothing -- the oracle stays
 * "parity unpinned" -- but it is the one independent witness the restatement in vof_oracle.c / pix2uv_oracle.c can get: the reference's
 * OWN kernel text executing on the MI355X.  This file only forwards plain-C arguments into the reference's entry
 * points (declared by their callers at ref src/oct_optical_flow.cc:12,15); it contains no reference code. */
#include <cstring>
#include <string>

#include "image.h"
#include "goesread.h"
#include "offlags.h"

void oct_variational_optical_flow(Image geo1i, Image geo2i, float *CTH, float *uarr, float *varr, int nx, int ny, int nc, OFFlags args);
void oct_pix2uv_cuda(GOESVar &goesData, double t2, float *uarr, float *varr, short *ur, short *vr, short *ur2, short *vr2, OFFlags args);

extern "C" int oct_refhip_vof(const float *img1, const float *img2, int nx, int ny, int nc, float *u, float *v, double alpha, double lambda,
                              double lambdac, double scaleF, double scsig, int kiters, int liters, int cgiters, int dozim, int device)
{
    Image a(nx, ny, nc), b(nx, ny, nc);
    a.data = const_cast<float *>(img1);
    b.data = const_cast<float *>(img2);
    OFFlags f = OFFlags();
    f.alpha = alpha; f.lambda = lambda; f.lambdac = lambdac; f.scaleF = scaleF; f.scsig = scsig;
    f.kiters = kiters; f.liters = liters; f.cgiters = cgiters; f.dozim = dozim; f.setdevice = device;
    oct_variational_optical_flow(a, b, nullptr, u, v, nx, ny, nc, f);
    return 0;
}

extern "C" int oct_refhip_pix2uv(double pph, double req, double rpol, double lam0, float xScale, float xOffset, float yScale, float yOffset,
                                 float g2xOffset, float g2yOffset, float lat1, float lon1, float lon0, float R, int minX, int minY, int nx, int ny,
                                 double t1, double t2, float *u, float *v, int pixuv, int mode, short *ur, short *vr, short *ur2, short *vr2, float *dT, int device)
{
    GOESVar g = GOESVar();
    g.nav.pph = pph; g.nav.req = req; g.nav.rpol = rpol; g.nav.lam0 = lam0;
    g.nav.xScale = xScale; g.nav.xOffset = xOffset; g.nav.yScale = yScale; g.nav.yOffset = yOffset;
    g.nav.g2xOffset = g2xOffset; g.nav.g2yOffset = g2yOffset; g.nav.lat1 = lat1; g.nav.lon1 = lon1; g.nav.lon0 = lon0; g.nav.R = R;
    g.nav.minX = minX; g.nav.minY = minY; g.nav.nx = nx; g.nav.ny = ny;
    g.t = t1;
    OFFlags f = OFFlags();
    f.pixuv = pixuv; f.dopolar = mode == 1; f.domerc = mode == 2; f.setdevice = device;
    oct_pix2uv_cuda(g, t2, u, v, ur, vr, ur2, vr2, f);
    if (dT) *dT = g.dT;
    return 0;
}
