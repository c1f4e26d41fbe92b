// Test tool for the NetCDF-4 layer (octane_amd/csrc/io).
//   io_demo --make-goes out.nc nx ny rad.bin t band xoff yoff   writes a GOES-R L1b look-alike: Rad (int16 counts from rad.bin,
//                                                            deflate-compressed), x, y, t, band_id, goes_imager_projection,
//                                                            planck_* and kappa0, with the attributes oct_goesread needs
//   io_demo --make-proj out.nc polar|merc nx ny rad.bin t xs xo ys yo lon lat1 R    a re-mapped image (float Rad)
//   io_demo --make-cth out.nc nx ny cth.bin                    cloud-top heights as CLAVR-x names them
//   io_demo --make-fg out.nc nx ny uv.bin                     a first-guess file: UFG, VFG (float32 from uv.bin)
//   io_demo --write-out out.nc GOES|POLAR|MERC nx ny nchan    runs oct_filewrite on a synthetic, fully populated GOESVar (no GPU)
//   io_demo --dump file.nc                                    one line per variable: name|shape|att=value;...
//   io_demo --read file.nc var type out.bin                   whole variable as short|int|float|double
//   io_demo --goesread file.nc                                oct_goesread on the file (channel 1, RAW, no navigation): prints rc=<code>; a file
//                                                            that is not a readable GOES-R L1b has to come back as an error, not a crash
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "nc4lite.hpp"
#include "../../include/octane_host.hpp"

using nc4lite::Type;

int oct_filewrite(std::string, std::string, GOESVar &, OFFlags);
int oct_goesread(std::string, std::string, int, int, GOESVar &, OFFlags &);

int main(int argc, char **argv)
{
    try {
        if ((argc == 10 || argc == 11) && !strcmp(argv[1], "--make-goes")) {
            const float smul = argc == 11 ? (float)atof(argv[10]) : 1.f;      // pixel size in units of 56 urad (2 km)
            const int nx = atoi(argv[3]), ny = atoi(argv[4]);
            std::vector<short> rad((size_t)nx * ny), x(nx), y(ny);
            FILE *f = fopen(argv[5], "rb");
            if (!f || fread(rad.data(), 2, rad.size(), f) != rad.size()) { printf("bad input\n"); return 2; }
            fclose(f);
            for (int i = 0; i < nx; i++) x[i] = (short)i;
            for (int j = 0; j < ny; j++) y[j] = (short)j;
            const double t = atof(argv[6]);
            const int band = atoi(argv[7]);
            nc4lite::Writer w(argv[2]);
            w.def_dim("y", ny); w.def_dim("x", nx); w.def_dim("band", 1);
            w.def_var("Rad", Type::Short, {"y", "x"}, 4);
            w.put_att("Rad", "scale_factor", 0.04572892f); w.put_att("Rad", "add_offset", -1.6443f);
            w.put_att("Rad", "long_name", std::string("ABI L1b Radiances"));
            w.def_var("x", Type::Short, {"x"});
            w.put_att("x", "scale_factor", 5.6e-05f * smul); w.put_att("x", "add_offset", (float)atof(argv[8]));
            w.def_var("y", Type::Short, {"y"});
            w.put_att("y", "scale_factor", -5.6e-05f * smul); w.put_att("y", "add_offset", (float)atof(argv[9]));
            w.def_var("t", Type::Double);
            w.put_att("t", "units", std::string("seconds since 2000-01-01 12:00:00"));
            w.def_var("band_id", Type::Byte, {"band"});
            w.def_var("goes_imager_projection", Type::Int);
            const char *gp = "goes_imager_projection";
            w.put_att(gp, "grid_mapping_name", std::string("geostationary"));
            w.put_att(gp, "perspective_point_height", 35786023.0);
            w.put_att(gp, "semi_major_axis", 6378137.0);
            w.put_att(gp, "semi_minor_axis", 6356752.31414);
            w.put_att(gp, "inverse_flattening", 298.2572221);
            w.put_att(gp, "latitude_of_projection_origin", 0.0);
            w.put_att(gp, "longitude_of_projection_origin", -75.0);
            w.put_att(gp, "sweep_angle_axis", std::string("x"));
            const float pl[5] = {10803.3f, 1392.74f, 0.07550f, 0.99975f, 0.0015839f};
            const char *pn[5] = {"planck_fk1", "planck_fk2", "planck_bc1", "planck_bc2", "kappa0"};
            for (int i = 0; i < 5; i++) { w.def_var(pn[i], Type::Float); w.put_var(pn[i], &pl[i]); }
            w.put_var("Rad", rad.data()); w.put_var("x", x.data()); w.put_var("y", y.data()); w.put_var("t", &t);
            w.put_var("band_id", &band);
            const int gip = -2147483647;
            w.put_var(gp, &gip);
            w.close();
            return 0;
        }
        if (argc == 15 && !strcmp(argv[1], "--make-proj")) {       // a re-mapped image as oct_polarread / oct_mercread expect it
            const bool polar = !strcmp(argv[3], "polar");
            const int nx = atoi(argv[4]), ny = atoi(argv[5]);
            std::vector<float> rad((size_t)nx * ny);
            std::vector<short> x(nx), y(ny);
            FILE *f = fopen(argv[6], "rb");
            if (!f || fread(rad.data(), 4, rad.size(), f) != rad.size()) { printf("bad input\n"); return 2; }
            fclose(f);
            for (int i = 0; i < nx; i++) x[i] = (short)i;
            for (int j = 0; j < ny; j++) y[j] = (short)j;
            const double t = atof(argv[7]);
            nc4lite::Writer w(argv[2]);
            w.def_dim("y", ny); w.def_dim("x", nx);
            w.def_var("Rad", Type::Float, {"y", "x"}, 4);
            w.def_var("x", Type::Short, {"x"});
            w.put_att("x", "scale_factor", (float)atof(argv[8])); w.put_att("x", "add_offset", (float)atof(argv[9]));
            w.def_var("y", Type::Short, {"y"});
            w.put_att("y", "scale_factor", (float)atof(argv[10])); w.put_att("y", "add_offset", (float)atof(argv[11]));
            w.def_var("t", Type::Double);
            w.put_att("t", "units", std::string("seconds since 2000-01-01 12:00:00"));
            w.def_var("grid_mapping", Type::Int);
            if (polar) { w.put_att("grid_mapping", "lon0", (float)atof(argv[12])); w.put_att("grid_mapping", "lat1", (float)atof(argv[13])); }
            else w.put_att("grid_mapping", "lon1", (float)atof(argv[12]));
            w.put_att("grid_mapping", "R", (float)atof(argv[14]));
            w.put_var("Rad", rad.data()); w.put_var("x", x.data()); w.put_var("y", y.data()); w.put_var("t", &t);
            const int gip = 7;
            w.put_var("grid_mapping", &gip);
            w.close();
            return 0;
        }
        if (argc == 6 && !strcmp(argv[1], "--make-cth")) {          // CLAVR-x look-alike: Cloud_Top_Height_Effective(ny, nx)
            const int nx = atoi(argv[3]), ny = atoi(argv[4]);
            std::vector<float> cth((size_t)nx * ny);
            FILE *f = fopen(argv[5], "rb");
            if (!f || fread(cth.data(), 4, cth.size(), f) != cth.size()) { printf("bad input\n"); return 2; }
            fclose(f);
            nc4lite::Writer w(argv[2]);
            w.def_dim("ny", ny); w.def_dim("nx", nx);
            w.def_var("Cloud_Top_Height_Effective", Type::Float, {"ny", "nx"}, 4);
            w.put_var("Cloud_Top_Height_Effective", cth.data());
            w.close();
            return 0;
        }
        if (argc == 6 && !strcmp(argv[1], "--make-fg")) {
            const int nx = atoi(argv[3]), ny = atoi(argv[4]);
            std::vector<float> uv((size_t)2 * nx * ny);
            FILE *f = fopen(argv[5], "rb");
            if (!f || fread(uv.data(), 4, uv.size(), f) != uv.size()) { printf("bad input\n"); return 2; }
            fclose(f);
            nc4lite::Writer w(argv[2]);
            w.def_dim("ny", ny); w.def_dim("nx", nx);
            w.def_var("UFG", Type::Float, {"ny", "nx"});
            w.def_var("VFG", Type::Float, {"ny", "nx"});
            w.put_var("UFG", uv.data()); w.put_var("VFG", uv.data() + (size_t)nx * ny);
            w.close();
            return 0;
        }
        if (argc == 7 && !strcmp(argv[1], "--write-out")) {
            const std::string ftype = argv[3];
            const int nx = atoi(argv[4]), ny = atoi(argv[5]), nc = atoi(argv[6]);
            const size_t n = (size_t)nx * ny;
            OFFlags args;
            octane_default_flags(args);
            args.ftype = ftype; args.oftype = 1; args.pixuv = 1; args.dosrsal = 1; args.outrad = true; args.putinterp = 0;
            args.doc2 = nc >= 2; args.doc3 = nc >= 3;
            GOESVar g;
            g.nav.nx = nx; g.nav.ny = ny;
            g.nav.xScale = 2000.f; g.nav.xOffset = -1000.f; g.nav.yScale = -2000.f; g.nav.yOffset = 3000.f;
            g.nav.lat1 = 70.f; g.nav.lon0 = -45.f; g.nav.lon1 = -100.f; g.nav.R = 6371228.f; g.nav.gipVal = 7.f;
            g.nav.radScale = 0.5f; g.nav.radOffset = -1.f; g.nav.g2xOffset = -1000.f; g.nav.g2yOffset = 3000.f;
            g.nav.pph = 35786023.f; g.nav.req = 6378137.f; g.nav.rpol = 6356752.5f; g.nav.inverse_flattening = 298.25f; g.nav.lat0 = 0.f; g.nav.lpo = -75.f;
            g.nav.fk1 = 1.f; g.nav.fk2 = 2.f; g.nav.bc1 = 3.f; g.nav.bc2 = 4.f; g.nav.kap1 = 5.f;
            std::vector<short> x(nx), y(ny), sv(n), sv2(n);
            std::vector<float> fu(n), fv(n), img(n * nc);
            for (int i = 0; i < nx; i++) x[i] = (short)i;
            for (int j = 0; j < ny; j++) y[j] = (short)j;
            for (size_t i = 0; i < n; i++) { fu[i] = 0.25f * (float)i; fv[i] = -0.5f * (float)i; sv[i] = (short)(i % 1000); sv2[i] = (short)(-(long)(i % 500)); }
            for (size_t i = 0; i < n * nc; i++) img[i] = (float)(i % 251);
            g.x = x.data(); g.y = y.data(); g.t = 7.1e8; g.tUnits = "seconds since 2000-01-01 12:00:00"; g.dT = 300.f;
            g.uPix = fu.data(); g.vPix = fv.data(); g.uVal = sv.data(); g.vVal = sv2.data(); g.uVal2 = sv.data(); g.vVal2 = sv2.data();
            g.dataSVal = sv.data();
            g.data.setdims(nx, ny, nc); g.data.data = img.data();
            const int rc = oct_filewrite(argv[2], ftype, g, args);
            g.data.data = nullptr;          // the vectors own the memory
            return rc;
        }
        if (argc == 3 && !strcmp(argv[1], "--goesread")) {
            GOESVar g;
            OFFlags args;
            octane_default_flags(args);
            const int rc = oct_goesread(argv[2], "RAW", 0, 1, g, args);
            printf("rc=%d\n", rc);
            return rc == 0 ? 0 : 4;
        }
        if (argc == 3 && !strcmp(argv[1], "--dump")) {
            fputs(nc4lite::describe(argv[2]).c_str(), stdout);
            return 0;
        }
        if (argc == 6 && !strcmp(argv[1], "--read")) {
            nc4lite::Reader r(argv[2]);
            size_t n = 1;
            for (size_t d : r.shape(argv[3])) n *= d;
            FILE *f = fopen(argv[5], "wb");
            const std::string ty = argv[4];
            if (ty == "short") { std::vector<short> v(n); r.read(argv[3], v.data()); fwrite(v.data(), 2, n, f); }
            else if (ty == "int") { std::vector<int> v(n); r.read(argv[3], v.data()); fwrite(v.data(), 4, n, f); }
            else if (ty == "float") { std::vector<float> v(n); r.read(argv[3], v.data()); fwrite(v.data(), 4, n, f); }
            else { std::vector<double> v(n); r.read(argv[3], v.data()); fwrite(v.data(), 8, n, f); }
            fclose(f);
            return 0;
        }
    } catch (const nc4lite::Error &e) {
        printf("error: %s\n", e.what());
        return 3;
    }
    printf("usage: see the header of tests/cpp/io_demo.cpp\n");
    return 1;
}
