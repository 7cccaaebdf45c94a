// TEMPORARY: symbols not yet implemented (sort, group-by) so the library links.
#include "common.h"
using namespace amdarrow;
extern "C" {
int amdarrow_sort_indices(amdarrow_ctx*, const amdarrow_array*, int, int, uint64_t*) { return SetError(AMDARROW_NOT_IMPLEMENTED, "sort_indices"); }
}
