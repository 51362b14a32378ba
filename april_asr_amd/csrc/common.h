// Shared small utilities: leveled logging (same env var and format family as the
// reference's src/log.h:23-56 / src/init.c:36-44), HIP error handling.
#pragma once
#include <cstdio>
#include <cstdlib>
#include <string>

namespace aprilx {

enum LogLevel { LOG_DEBUG = 0, LOG_INFO = 1, LOG_WARNING = 2, LOG_ERROR = 3, LOG_NONE = 4 };
extern int g_loglevel;

#define APX_LOG(level, tag, fmt, ...)                                                             \
    do {                                                                                          \
        if ((level) >= ::aprilx::g_loglevel)                                                      \
            fprintf(stderr, "libapril(mi355x): (%s:%d) [%s] " fmt "\n", __FILE__, __LINE__, tag,  \
                    ##__VA_ARGS__);                                                               \
    } while (0)
#define LOGD(fmt, ...) APX_LOG(::aprilx::LOG_DEBUG, "DEBUG", fmt, ##__VA_ARGS__)
#define LOGI(fmt, ...) APX_LOG(::aprilx::LOG_INFO, "INFO", fmt, ##__VA_ARGS__)
#define LOGW(fmt, ...) APX_LOG(::aprilx::LOG_WARNING, "WARNING", fmt, ##__VA_ARGS__)
#define LOGE(fmt, ...) APX_LOG(::aprilx::LOG_ERROR, "ERROR", fmt, ##__VA_ARGS__)

// Unrecoverable backend error => log + abort(), the reference's convention for
// ORT failures (src/ort_util.h:29-38).
#define HIP_CHECK(expr)                                                                \
    do {                                                                               \
        hipError_t e_ = (expr);                                                        \
        if (e_ != hipSuccess) {                                                        \
            LOGE("HIP: %s failed: %s", #expr, hipGetErrorString(e_));                  \
            abort();                                                                   \
        }                                                                              \
    } while (0)

}  // namespace aprilx
