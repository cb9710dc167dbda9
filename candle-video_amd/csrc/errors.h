// Error plumbing shared by device and host-only translation units: thread-local last error, C-ABI returns int codes.
#pragma once
#include <string>
enum { LTX_OK = 0, LTX_ERR_ARG = 1, LTX_ERR_HIP = 2, LTX_ERR_MISSING_WEIGHT = 3, LTX_ERR_UNSUPPORTED = 4 };
void ltx_set_error(const std::string& s);
#define LTX_FAIL(code, msg) do { ltx_set_error(std::string(msg)); return (code); } while (0)
#define LTX_TRY(expr) do { int _rc = (expr); if (_rc != LTX_OK) return _rc; } while (0)
