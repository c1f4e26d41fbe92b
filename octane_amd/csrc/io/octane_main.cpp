// octane_main.cpp -- the `octane` command line on this library: read two GOES-R L1b files, compute the flow, write
// outfile.nc.  Control flow of ref src/main.cc:110-484 for the file types this build reads (GOES fixed grid, one
// channel, optional first guess); the flags, their defaults and quirks come from octane_parse_command_line
// (host_shim.cpp), the work from oct_optical_flow (same file).
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
                     "Flag spellings and defaults are the reference's (src/main.cc:42-108).\n";
        return 0;
    }
    if (args.doCTH == 1 || args.dopolar == 1 || args.domerc == 1 || args.doahi == 1 || args.doc2 == 1 || args.doc3 == 1 || args.dointerp == 1) {
        std::cout << "This build reads GOES fixed-grid L1b files with one channel; -Polar/-Merc/-ahi, cloud-top heights, "
                     "extra channels and -interp need the reference's readers, exiting\n";
        return 0;
    }
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
    goesData.nav.g2xOffset = goesData2.nav.xOffset;                        // ref main.cc:399-403
    goesData.nav.g2yOffset = goesData2.nav.yOffset;
    if (args.dofirstguess == 1) oct_fileread(c.f1fg, "FIRSTGUESS", "RAW", 0, 0, goesData, args);
    oct_optical_flow(goesData, goesData2, args);
    args.putinterp = 0;
    const std::string outname = c.outdir + "outfile.nc";                   // ref main.cc:443
    const int rc = oct_filewrite(outname, args.ftype, goesData, args);
    if (rc != 0) return 1;
    std::cout << outname << " written\n";
    std::cout << "OCTANE completed, exiting\n";
    return 0;
}
