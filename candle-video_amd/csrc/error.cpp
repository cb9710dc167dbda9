#include "errors.h"
static thread_local std::string g_last_error;
void ltx_set_error(const std::string& s) { g_last_error = s; }
extern "C" const char* ltx_last_error(void) { return g_last_error.c_str(); }
