// Host program written the way OCTANE's own main.cc drives the path: parse the `octane` command line, fill two
// GOESVar objects, call oct_optical_flow(), read the outputs.  Test driver:
//   host_demo --parse-only <octane args...>          prints the parsed OFFlags as key=value lines (no GPU)
//   host_demo --run nx ny in.bin out.bin <octane args...>
//        in.bin : img1, img2 (float32, nx*ny each);  out.bin : uPix, vPix (float32), uVal, vVal, uVal2, vVal2 (int16)
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "octane_host.hpp"

static void dump(const OctaneCommandLine &c)
{
    const OFFlags &a = c.args;
    printf("help=%d\nf1=%s\nf2=%s\nf1c=%s\nfc21=%s\nfc22=%s\nf1fg=%s\noutdir=%s\ninterploc=%s\nftype=%s\n", c.show_help, c.f1.c_str(), c.f2.c_str(),
           c.f1c.c_str(), c.fc21.c_str(), c.fc22.c_str(), c.f1fg.c_str(), c.outdir.c_str(), c.interploc.c_str(), a.ftype.c_str());
    printf("alpha=%.17g\nlambda=%.17g\nlambdac=%.17g\nscsig=%.17g\nscaleF=%.17g\nalpha2=%.17g\n", a.alpha, a.lambda, a.lambdac, a.scsig, a.scaleF, a.alpha2);
    printf("kiters=%d\nliters=%d\ncgiters=%d\ndozim=%d\nsetdevice=%d\npixuv=%d\ndopolar=%d\ndomerc=%d\ndososm=%d\ndosrsal=%d\n", a.kiters, a.liters,
           a.cgiters, a.dozim, a.setdevice, a.pixuv, a.dopolar, a.domerc, a.dososm, a.dosrsal);
    printf("doCTH=%d\ndoc2=%d\ndoc3=%d\ndofirstguess=%d\noftype=%d\ndocorn=%d\nir=%d\nrad=%d\nsrad=%d\ndeltat=%g\ninterpcth=%d\n", a.doCTH, a.doc2, a.doc3,
           a.dofirstguess, a.oftype, a.docorn, a.ir, a.rad, a.srad, a.deltat, a.interpcth);
    printf("outnav=%d\noutraw=%d\noutrad=%d\noutctp=%d\nsetNormMax=%d\nNormMax=%g\nsetNormMin2=%d\nNormMin2=%g\n", a.outnav, a.outraw, a.outrad, a.outctp,
           a.setNormMax, a.NormMax, a.setNormMin2, a.NormMin2);
}

int main(int argc, char **argv)
{
    if (argc >= 2 && !strcmp(argv[1], "--parse-only")) {
        std::vector<const char *> av; av.push_back("octane");
        for (int i = 2; i < argc; i++) av.push_back(argv[i]);
        dump(octane_parse_command_line((int)av.size(), av.data()));
        return 0;
    }
    if (argc >= 6 && !strcmp(argv[1], "--run")) {
        const int nx = atoi(argv[2]), ny = atoi(argv[3]);
        const long n = (long)nx * ny;
        std::vector<const char *> av; av.push_back("octane");
        for (int i = 6; i < argc; i++) av.push_back(argv[i]);
        OctaneCommandLine c = octane_parse_command_line((int)av.size(), av.data());
        std::vector<float> a(n), b(n);
        FILE *f = fopen(argv[4], "rb");
        if (!f || fread(a.data(), 4, n, f) != (size_t)n || fread(b.data(), 4, n, f) != (size_t)n) { printf("bad input\n"); return 2; }
        fclose(f);
        GOESVar g1 = GOESVar(), g2 = GOESVar();
        g1.data = Image(nx, ny, 1); g1.data.data = a.data();
        g2.data = Image(nx, ny, 1); g2.data.data = b.data();
        g1.nav = GOESNAVVar(); g2.nav = GOESNAVVar();
        g1.nav.nx = nx; g1.nav.ny = ny; g2.nav.nx = nx; g2.nav.ny = ny;
        // GOES-16 CONUS-like fixed grid, same sector for both images
        g1.nav.pph = 35786023.0; g1.nav.req = 6378137.0; g1.nav.rpol = 6356752.31414; g1.nav.lam0 = -1.308996939;
        g1.nav.xScale = 5.6e-05f; g1.nav.xOffset = -0.101332f; g1.nav.yScale = -5.6e-05f; g1.nav.yOffset = 0.128212f;
        g1.nav.g2xOffset = g1.nav.xOffset; g1.nav.g2yOffset = g1.nav.yOffset; g1.nav.minX = 0; g1.nav.minY = 0;
        g1.t = 1000.0; g2.t = 1300.0;
        int rc = oct_optical_flow(g1, g2, c.args);
        f = fopen(argv[5], "wb");
        fwrite(g1.uPix, 4, n, f); fwrite(g1.vPix, 4, n, f);
        fwrite(g1.uVal, 2, n, f); fwrite(g1.vVal, 2, n, f); fwrite(g1.uVal2, 2, n, f); fwrite(g1.vVal2, 2, n, f);
        fclose(f);
        printf("rc=%d dT=%g\n", rc, g1.dT);
        return rc == 1 ? 0 : 1;
    }
    printf("usage: host_demo --parse-only ... | --run nx ny in.bin out.bin ...\n");
    return 2;
}
