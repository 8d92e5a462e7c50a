/* test-only stand-in: see ../hdk_mock.h.  The real header brings OpenGL with it; so does this one (the system's core-profile header when
 * there is one, else the handful of entry points the scene hook uses). */
#include "../hdk_mock.h"
#if __has_include(<GL/glcorearb.h>)
#define GL_GLEXT_PROTOTYPES 1
#include <GL/glcorearb.h>
#else
#include "gl_min.h"
#endif
