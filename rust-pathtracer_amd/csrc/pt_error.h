// pt_error.h — the thread-local message behind pt_last_error(), shared by the translation units of libptamd.so
#ifndef PT_ERROR_H
#define PT_ERROR_H
#include <string>
void pt_set_error(const std::string& message);
#endif
