/* Declaration-only stand-in for Biostrings' C interface: see README.md in this directory (test infrastructure). */
#ifndef RGLUE_STUB_BIOSTRINGS_H
#define RGLUE_STUB_BIOSTRINGS_H
typedef struct { const char* ptr; int length; } Chars_holder;
typedef struct { int opaque; } XStringSet_holder;
XStringSet_holder hold_XStringSet(SEXP);
int get_length_from_XStringSet_holder(const XStringSet_holder*);
Chars_holder get_elt_from_XStringSet_holder(const XStringSet_holder*, int);
char DNAdecode(char);
#endif
