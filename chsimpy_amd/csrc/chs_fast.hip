// placeholder, replaced below
#include "chs_common.h"
bool chs_fast_supported(int N, int dtype) { (void)N; (void)dtype; return false; }
int chs_fast_init(Engine* E) { (void)E; chs_set_error("fast engine not built"); return CHS_EINVAL; }
void chs_fast_free(Engine* E) { (void)E; }
int chs_fast_dct2d(Engine* E, const void* in, void* out, bool inverse) { (void)E; (void)in; (void)out; (void)inverse; return CHS_EINVAL; }
int chs_fast_enter(Engine* E) { (void)E; return CHS_EINVAL; }
int chs_fast_step(Engine* E) { (void)E; return CHS_EINVAL; }
