// Declaration-only stand-in for <Rcpp.h>: see README.md in this directory (test infrastructure, nothing is built from it).
#ifndef RGLUE_STUB_RCPP_H
#define RGLUE_STUB_RCPP_H
#include <cstddef>
#include <exception>
#include <string>
#include <type_traits>

struct SEXPREC;
typedef SEXPREC* SEXP;
typedef std::ptrdiff_t R_xlen_t;
extern SEXP R_NilValue;
const char* CHAR(SEXP);
SEXP STRING_ELT(SEXP, R_xlen_t);
int Rf_length(SEXP);
SEXP rglue_stub_error(const char*);

#define BEGIN_RCPP try {
#define END_RCPP } catch (std::exception& e__) { return rglue_stub_error(e__.what()); } return R_NilValue;

namespace Rcpp {

class RObject {
public:
    RObject(SEXP);
    bool isS4() const;
    bool hasAttribute(const char*) const;
    operator SEXP() const;
};

class String {
public:
    String(const std::string&);
    String(const char*);
    operator SEXP() const;
};

template <typename T> T as(SEXP);

// an element of a generic vector / list: readable as SEXP, assignable from anything that converts to SEXP or a string
class Proxy {
public:
    operator SEXP() const;
    operator RObject() const;
    template <typename U> Proxy& operator=(const U&);
};

template <typename T>
class Vector : public RObject {
public:
    Vector();
    Vector(SEXP);
    Vector(const Proxy&);
    template <typename N, typename = typename std::enable_if<std::is_integral<N>::value>::type> explicit Vector(N n);   // (as Rcpp: an integral size is an exact match, never a null SEXP)
    template <typename It> Vector(It first, It last);
    R_xlen_t size() const;
    T* begin();
    T* end();
    T& operator[](R_xlen_t);
    const T& operator[](R_xlen_t) const;
    class StringVectorStub names() const;
};
typedef Vector<double> NumericVector;
typedef Vector<int> IntegerVector;
typedef Vector<int> LogicalVector;

class StringVectorStub : public RObject {
public:
    StringVectorStub();
    StringVectorStub(SEXP);
    StringVectorStub(const RObject&);
    template <typename N, typename = typename std::enable_if<std::is_integral<N>::value>::type> explicit StringVectorStub(N n);
    R_xlen_t size() const;
    Proxy operator[](R_xlen_t);
};
typedef StringVectorStub StringVector;

class List : public RObject {
public:
    List();
    List(SEXP);
    template <typename N, typename = typename std::enable_if<std::is_integral<N>::value>::type> explicit List(N n);
    R_xlen_t size() const;
    Proxy operator[](R_xlen_t);
    template <typename... Args> static List create(const Args&...);
};

template <> std::string as<std::string>(SEXP);

}  // namespace Rcpp
#endif
