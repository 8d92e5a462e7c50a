/* test-only stub of the ONE class of the reference's include/GEO_GSplat.h that the glue names (the reference's header itself is
 * not pulled in: it needs the real GEO library) */
#include "hdk_mock.h"
class GEO_PrimGsplat : public GEO_Primitive {};
