// Declaration-only stand-in for the reference package's src/utils.h (it stays in the shimmed package): the four scalar
// checkers the shims call.  See README.md in this directory (test infrastructure).
#ifndef RGLUE_STUB_UTILS_H
#define RGLUE_STUB_UTILS_H
#include "Rcpp.h"
#include <string>
int check_integer_scalar(Rcpp::RObject, const char*);
double check_numeric_scalar(Rcpp::RObject, const char*);
bool check_logical_scalar(Rcpp::RObject, const char*);
std::string check_string(Rcpp::RObject, const char*);
#endif
