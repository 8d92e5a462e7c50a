/* test-only stand-in: see ../hdk_mock.h */
#include "../hdk_mock.h"
