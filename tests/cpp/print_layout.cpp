// Prints sizeof / offsetof of the host boundary types.  Compiled twice by tests/test_host_abi.py: once against
// the reference's own headers (where /root/reference exists) and once against include/ of this repo; the two
// outputs must be identical for the drop-in claim to hold.
#include <cstddef>
#include <cstdio>
#include <string>
#include <type_traits>
#include "image.h"
#include "goesread.h"
#include "offlags.h"

#define P(T, m) printf(#T "." #m " %zu %zu\n", offsetof(T, m), sizeof(((T *)0)->m))
#pragma GCC diagnostic ignored "-Winvalid-offsetof"
int main()
{
    printf("sizeof Image %zu OFFlags %zu GOESNAVVar %zu GOESVar %zu\n", sizeof(Image), sizeof(OFFlags), sizeof(GOESNAVVar), sizeof(GOESVar));
    // what decides HOW an aggregate travels by value (registers / stack copy against a hidden reference to a temporary)
    printf("trivially_copyable Image %d OFFlags %d GOESNAVVar %d GOESVar %d  trivially_destructible Image %d OFFlags %d\n",
           (int)std::is_trivially_copyable<Image>::value, (int)std::is_trivially_copyable<OFFlags>::value, (int)std::is_trivially_copyable<GOESNAVVar>::value,
           (int)std::is_trivially_copyable<GOESVar>::value, (int)std::is_trivially_destructible<Image>::value, (int)std::is_trivially_destructible<OFFlags>::value);
    P(Image, data); P(Image, nrow); P(Image, ncol); P(Image, nchannels);
    P(OFFlags, farn); P(OFFlags, pixuv); P(OFFlags, dopolar); P(OFFlags, domerc); P(OFFlags, doahi); P(OFFlags, dosrsal);
    P(OFFlags, dososm); P(OFFlags, dofirstguess); P(OFFlags, ftype); P(OFFlags, dointerp); P(OFFlags, docorn);
    P(OFFlags, putinterp); P(OFFlags, interpcth); P(OFFlags, doinv); P(OFFlags, doctt); P(OFFlags, dozim); P(OFFlags, oftype);
    P(OFFlags, doc2); P(OFFlags, doc3); P(OFFlags, ir); P(OFFlags, rad); P(OFFlags, srad); P(OFFlags, setdevice);
    P(OFFlags, fpyr_scale); P(OFFlags, flevels); P(OFFlags, fwinsize); P(OFFlags, fiterations); P(OFFlags, poly_n);
    P(OFFlags, poly_sigma); P(OFFlags, deltat); P(OFFlags, uif); P(OFFlags, fg); P(OFFlags, doCTH);
    P(OFFlags, lambda); P(OFFlags, alpha); P(OFFlags, alpha2); P(OFFlags, lambdac); P(OFFlags, scsig); P(OFFlags, filtsigma);
    P(OFFlags, scaleF); P(OFFlags, kiters); P(OFFlags, liters); P(OFFlags, cgiters); P(OFFlags, miters); P(OFFlags, setnorms);
    P(OFFlags, NormMax); P(OFFlags, NormMin); P(OFFlags, NormMax2); P(OFFlags, NormMin2); P(OFFlags, NormMax3); P(OFFlags, NormMin3);
    P(OFFlags, outnav); P(OFFlags, outraw); P(OFFlags, outrad); P(OFFlags, outctp);
    P(OFFlags, setNormMax); P(OFFlags, setNormMin); P(OFFlags, setNormMax2); P(OFFlags, setNormMin2); P(OFFlags, setNormMax3); P(OFFlags, setNormMin3);
    P(GOESNAVVar, pph); P(GOESNAVVar, req); P(GOESNAVVar, rpol); P(GOESNAVVar, lam0); P(GOESNAVVar, inverse_flattening); P(GOESNAVVar, lat0);
    P(GOESNAVVar, gipVal); P(GOESNAVVar, xScale); P(GOESNAVVar, xOffset); P(GOESNAVVar, yScale); P(GOESNAVVar, yOffset);
    P(GOESNAVVar, g2xOffset); P(GOESNAVVar, g2yOffset); P(GOESNAVVar, radOffset3); P(GOESNAVVar, nx2); P(GOESNAVVar, nx); P(GOESNAVVar, ny);
    P(GOESNAVVar, CTHy); P(GOESNAVVar, minXc); P(GOESNAVVar, minX); P(GOESNAVVar, minY); P(GOESNAVVar, maxY);
    P(GOESNAVVar, lat1); P(GOESNAVVar, lon1); P(GOESNAVVar, lon0); P(GOESNAVVar, R);
    P(GOESVar, latVal); P(GOESVar, CTP); P(GOESVar, CTI); P(GOESVar, dataVal3); P(GOESVar, data); P(GOESVar, occlusion);
    P(GOESVar, uVal); P(GOESVar, vVal); P(GOESVar, uVal2); P(GOESVar, vVal2); P(GOESVar, uPix); P(GOESVar, vPix);
    P(GOESVar, UFG); P(GOESVar, accel); P(GOESVar, CTHVal); P(GOESVar, CTHInv); P(GOESVar, dataSValfloat3);
    P(GOESVar, t); P(GOESVar, tint); P(GOESVar, dT); P(GOESVar, frdt); P(GOESVar, band3); P(GOESVar, nav); P(GOESVar, tUnits);
    return 0;
}
