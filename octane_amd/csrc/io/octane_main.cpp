// octane_main.cpp -- the `octane` command line on this library: read two images (GOES-R L1b, polar or Mercator
// re-mapped; up to three channels; optional cloud-top heights and first guess), compute the flow, write outfile*.nc.
// Control flow of ref src/main.cc:110-484 without -ahi and the -interp frames; the flags, their defaults and quirks come
// from octane_parse_command_line (host_shim.cpp), the work from oct_optical_flow (same file).
#include <cstdio>
#include <iostream>
#include <string>

#include "../../../include/octane_host.hpp"

int oct_fileread(std::string, std::string, std::string, int, int, GOESVar &, OFFlags &);
int oct_filewrite(std::string, std::string, GOESVar &, OFFlags);

int main(int argc, char **argv)
{
    OctaneCommandLine c = octane_parse_command_line(argc, argv);
    OFFlags &args = c.args;
    if (c.show_help) {                                     // ref main.cc:112-165 prints the full option list
        std::cout << "usage: octane -i1 <file 1> -i2 <file 2> [-alpha a] [-lambda l] [-kiters k] [-liters l] [-o outdir/]\n"
                     "              [-sosm [-rad r] [-srad s]] [-brox] [-pd] [-srsal] [-firstguess file] [-set_device n]\n"
                     "              [-Polar | -Merc] [-ic21 f -ic22 f [-ic31 f -ic32 f]] [-i1cth file]\n"
                     "Flag spellings and defaults are the reference's (src/main.cc:42-108).\n";
        return 0;
    }
    if (args.doahi == 1 || args.dointerp == 1) {
        std::cout << "This build has no -ahi reader and does not write the frames of -interp, exiting\n";
        return 0;
    }
    if (args.doc2 == 1 && (c.fc22.empty() || c.fc22 == "none")) { printf("Missing files for second channel...stopping \n"); return 0; }      // ref main.cc:352-361
    if (args.doc3 == 1 && (c.fc32.empty() || c.fc32 == "none")) { printf("Missing files for third channel...stopping \n"); return 0; }
    GOESVar goesData, goesData2;
    std::cout << "Here are the file names being used: \n";
    std::cout << "File 1 : " << c.f1 << std::endl;
    std::cout << "File 2 : " << c.f2 << std::endl;
    oct_fileread(c.f1, args.ftype, "RAW", 1, 1, goesData, args);           // ref main.cc:396-397
    oct_fileread(c.f2, args.ftype, "RAW", 0, 1, goesData2, args);
    if (goesData.nav.nx != goesData2.nav.nx || goesData.nav.ny != goesData2.nav.ny) {
        std::cout << "The two images differ in size, exiting\n";
        return 0;
    }
    // ref main.cc:399-403 sets these for GOES files only and leaves them uninitialised otherwise, although the
    // sector-moved guard of oct_pix2uv_cuda reads them whatever the projection (p2u:295); set for every file type here
    goesData.nav.g2xOffset = goesData2.nav.xOffset;
    goesData.nav.g2yOffset = goesData2.nav.yOffset;
    if (args.doCTH == 1) oct_fileread(c.f1c, "CLAVRX", "RAW", 0, 0, goesData, args);                  // ref main.cc:405-410
    if (args.dofirstguess == 1) oct_fileread(c.f1fg, "FIRSTGUESS", "RAW", 0, 0, goesData, args);
    for (int ch = 2; ch <= 3; ch++) {                                      // ref main.cc:413-435
        if ((ch == 2 ? args.doc2 : args.doc3) != 1) continue;
        if (args.domerc == 1) {
            printf("Mercator multi-channel not compatable with this version, use single channel only\n");
            return 0;
        }
        oct_fileread(ch == 2 ? c.fc21 : c.fc31, args.ftype, "RAW", 1, ch, goesData, args);
        oct_fileread(ch == 2 ? c.fc22 : c.fc32, args.ftype, "RAW", 0, ch, goesData2, args);
    }
    oct_optical_flow(goesData, goesData2, args);
    args.putinterp = 0;
    std::string outname = c.outdir + "outfile.nc";                         // ref main.cc:443-445
    if (args.ftype == "POLAR") outname = c.outdir + "outfile_polar.nc";
    if (args.ftype == "MERC") outname = c.outdir + "outfile_merc.nc";
    const int rc = oct_filewrite(outname, args.ftype, goesData, args);
    if (rc != 0) return 1;
    std::cout << outname << " written\n";
    std::cout << "OCTANE completed, exiting\n";
    return 0;
}
