// Driver for tests/test_ref_caller_link.py: main() for the REFERENCE's own caller object.  This file is compiled against the
// reference's headers (-I /root/reference/include) and linked with the object made from /root/reference/src/oct_optical_flow.cc
// where it lies; liboctane_host.so supplies what that object calls (oct_variational_optical_flow, oct_pix2uv_cuda, oct_uv2pix,
// oct_srsal_cu, oct_patch_match_optical_flow -- declared by the reference at src/oct_optical_flow.cc:11-17).  The oct_optical_flow()
// called below is the reference's (the executable's own definition wins over the library's).
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>
#include "image.h"
#include "goesread.h"
#include "offlags.h"

int oct_optical_flow(GOESVar &, GOESVar &, OFFlags &);

int main(int argc, char **argv)
{
    const int nx = argc > 1 ? atoi(argv[1]) : 64, ny = argc > 2 ? atoi(argv[2]) : 48;
    const long n = (long)nx * ny;
    std::vector<float> a(n), b(n);
    for (long i = 0; i < n; i++) { a[i] = (float)((i * 37) % 251); b[i] = (float)((i * 41 + 7) % 241); }
    a[0] = 11.5f; b[0] = 22.25f;
    OFFlags args;                          // ref src/main.cc:53-108 (the fields the path reads)
    args.farn = 0; args.pixuv = 0; args.dosrsal = 0; args.dopolar = 0; args.domerc = 0; args.ftype = "GOES";
    args.dofirstguess = 0; args.ir = 0; args.dososm = 0; args.dointerp = 0; args.docorn = 0; args.rad = 2; args.srad = 2;
    args.lambda = 1.25; args.alpha = 5.5; args.scaleF = 0.5; args.kiters = 3; args.lambdac = 0.125; args.liters = 2; args.cgiters = 7;
    args.scsig = 400.; args.doc2 = 0; args.doc3 = 0; args.doCTH = 0; args.dozim = 1; args.setdevice = 0;
    GOESVar g1 = GOESVar(), g2 = GOESVar();
    g1.data = Image(nx, ny, 1); g1.data.data = a.data();
    g2.data = Image(nx, ny, 1); g2.data.data = b.data();
    g1.nav = GOESNAVVar(); g2.nav = GOESNAVVar();
    g1.nav.nx = nx; g1.nav.ny = ny; g2.nav.nx = nx; g2.nav.ny = ny;
    g1.nav.pph = 35786023.0; g1.nav.req = 6378137.0; g1.nav.rpol = 6356752.31414; g1.nav.lam0 = -1.308996939;
    g1.nav.xScale = 5.6e-05f; g1.nav.xOffset = -0.101332f; g1.nav.yScale = -5.6e-05f; g1.nav.yOffset = 0.128212f;
    g1.nav.g2xOffset = g1.nav.xOffset; g1.nav.g2yOffset = g1.nav.yOffset; g1.nav.minX = 0; g1.nav.minY = 0;
    g1.t = 1000.0; g2.t = 1300.0;
    const int rc = oct_optical_flow(g1, g2, args);
    double su = 0., sv = 0.;
    for (long i = 0; i < n; i++) { su += g1.uPix[i]; sv += g1.vPix[i]; }
    printf("rc=%d dT=%g sum_u=%.9g sum_v=%.9g uVal0=%d\n", rc, g1.dT, su, sv, (int)g1.uVal[0]);
    return rc == 1 ? 0 : 1;
}
