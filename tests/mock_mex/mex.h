/* Declaration-only stand-in for MATLAB's mex.h (R2018a interleaved-complex API), TEST USE ONLY:
 * tests/test_mex_gateway.py compiles matlab/vohip_mex.cpp against it so that the gateway's calls are
 * type-checked against include/vohip.h in an environment without MATLAB.  Nothing here is defined;
 * the real header and libmx / libmex come with MATLAB. */
#ifndef MOCK_MEX_H
#define MOCK_MEX_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif
typedef struct mxArray_tag mxArray;
typedef size_t mwSize;
typedef bool mxLogical;
typedef float mxSingle;
typedef double mxDouble;
typedef uint8_t mxUint8;
typedef enum { mxREAL = 0, mxCOMPLEX = 1 } mxComplexity;
typedef enum { mxDOUBLE_CLASS = 6, mxSINGLE_CLASS = 7, mxUINT8_CLASS = 9 } mxClassID;

void mexErrMsgIdAndTxt(const char *id, const char *fmt, ...) __attribute__((noreturn));
int mexAtExit(void (*fn)(void));
void mexLock(void);

bool mxIsDouble(const mxArray *a);
bool mxIsSingle(const mxArray *a);
bool mxIsUint8(const mxArray *a);
bool mxIsComplex(const mxArray *a);
bool mxIsChar(const mxArray *a);
bool mxIsStruct(const mxArray *a);
bool mxIsEmpty(const mxArray *a);
bool mxIsCell(const mxArray *a);
mxArray *mxGetCell(const mxArray *a, size_t index);
mwSize mxGetNumberOfDimensions(const mxArray *a);
const mwSize *mxGetDimensions(const mxArray *a);
size_t mxGetNumberOfElements(const mxArray *a);
size_t mxGetM(const mxArray *a);
size_t mxGetN(const mxArray *a);
double mxGetScalar(const mxArray *a);
int mxGetString(const mxArray *a, char *buf, mwSize buflen);
mxDouble *mxGetDoubles(const mxArray *a);
mxSingle *mxGetSingles(const mxArray *a);
mxUint8 *mxGetUint8s(const mxArray *a);
mxLogical *mxGetLogicals(const mxArray *a);
mxArray *mxGetField(const mxArray *a, size_t index, const char *name);
void mxSetField(mxArray *a, size_t index, const char *name, mxArray *value);

mxArray *mxCreateDoubleMatrix(mwSize m, mwSize n, mxComplexity flag);
mxArray *mxCreateDoubleScalar(double v);
mxArray *mxCreateLogicalMatrix(mwSize m, mwSize n);
mxArray *mxCreateLogicalScalar(bool v);
mxArray *mxCreateNumericMatrix(mwSize m, mwSize n, mxClassID cls, mxComplexity flag);
mxArray *mxCreateNumericArray(mwSize ndim, const mwSize *dims, mxClassID cls, mxComplexity flag);
mxArray *mxCreateStructMatrix(mwSize m, mwSize n, int nfields, const char **names);
mxArray *mxDuplicateArray(const mxArray *a);
void mxDestroyArray(mxArray *a);
void *mxCalloc(size_t n, size_t size);
void mxFree(void *p);

void mexFunction(int nlhs, mxArray *plhs[], int nrhs, const mxArray *prhs[]);
#ifdef __cplusplus
}
#endif
#endif
