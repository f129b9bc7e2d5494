// Error reporting and version for libsamx.
#include "samx_common.h"
#include "../../include/samx.h"
#include <stdarg.h>

namespace samx {

char* err_buf() {
    static thread_local char buf[512] = {0};
    return buf;
}

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(err_buf(), 512, fmt, ap);
    va_end(ap);
    return code;
}

}  // namespace samx

extern "C" {

const char* samx_last_error(void) { return samx::err_buf(); }
int samx_version(void) { return 100; }

int samx_stream_synchronize(void* stream) {
    SAMX_CHECK_HIP(hipStreamSynchronize((hipStream_t)stream));
    return SAMX_OK;
}

int64_t samx_mlp_param_count(const samx_mlp* d) {
    if (!d) return -1;
    int64_t ln = d->layernorm ? 2 * (int64_t)(d->h1 + d->h2) : 0;
    return (int64_t)d->in_dim * d->h1 + d->h1 + (int64_t)d->h1 * d->h2 + d->h2 +
           (int64_t)d->h2 * d->out_dim + d->out_dim + ln;
}

}  // extern "C"
