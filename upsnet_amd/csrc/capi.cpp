// capi.cpp -- error reporting + ABI version of libupsnet_hip.so.
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include "upsnet_hip.h"

static thread_local char g_err[512] = "";

int ups_set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return 1;
}

extern "C" const char *upsnet_last_error(void) { return g_err; }

static thread_local char g_form[96] = "";

void ups_set_form(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_form, sizeof(g_form), fmt, ap);
    va_end(ap);
}

extern "C" const char *upsnet_last_kernel_form(void) { return g_form; }
extern "C" int upsnet_abi_version(void) { return 1; }
