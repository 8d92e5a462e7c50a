/* GR_GSplat.h -- shadows the reference's include/GR_GSplat.h (hdk/build.sh puts this directory first on the include
 * path), so that the reference's unchanged src/GEO_GSplat.C (`#include "GR_GSplat.h"`, :13, where it registers
 * GR_PrimGsplatHook, :494-498) picks up the hook class of hdk/GR_GSplat_hip.h.  NOT COMPILED IN THIS REPOSITORY. */
#include "GR_GSplat_hip.h"
