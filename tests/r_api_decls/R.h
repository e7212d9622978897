/* TYPE-CHECK HARNESS, NOT R -- see Rinternals.h in this directory. */
#include <stdint.h>
