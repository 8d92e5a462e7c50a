/*
 * gsplat_plugin_hip.C -- unity translation unit of the plugin over libgsplat_hip (the reference's
 * /root/reference/gsplat_plugin/gsplat_plugin.C includes its seven sources the same way for hcustom).
 * From the reference tree (unchanged): the custom primitive, the SOP, the logger.  From this repo: the two glue files.
 * NOT COMPILED IN THIS REPOSITORY (needs $HFS): hdk/build.sh.
 */
#include "src/GEO_GSplat.C"         /* reference, unchanged: newGeometryPrim, registers GR_PrimGsplatHook */
#include "src/SOP_GSplat.C"         /* reference, unchanged: newSopOperator */
#include "src/GSplatLogger.C"       /* reference, unchanged */
#include "GR_GSplat_hip.C"          /* this repo: hdk/ */
#include "DM_GSplatHook_hip.C"      /* this repo: hdk/ -- newRenderHook */
