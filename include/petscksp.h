/*
 * petscksp.h -- the PETSc Mat/Vec/KSP call surface that SyamVangara/multigrid-petsc uses, implemented by
 * libmgpetsc.so on top of the MI355X kernel ABI (include/mgk.h).
 *
 * This header is the drop-in boundary of the product: the reference's own
 * src/{poisson,solver,matbuild,mesh,problem,array}.c include it as <petscksp.h> / "petscksp.h"
 * (include/header.h:12, include/solver.h:14, include/mesh.h:11) and link against libmgpetsc.so instead
 * of ${PETSC_KSP_LIB} (makefile:38).  It is NOT PETSc and carries none of its source: only the
 * symbols the reference references, with PETSc's C signatures.
 *
 * Scope.  Fully implemented on the GPU: everything `-cycle 0` (MultigridVcycle, src/solver.c:1414-1575)
 * touches, `-cycle 8` (MultigridPetscPCMG, src/solver.c:1884-1989: PCMG with richardson / chebyshev / preonly level
 * solvers, jacobi / none / lu preconditioners, exact coarse solve) and `-cycle 1` (MultigridIcycle, src/solver.c:1991-2060).
 * Objects of the other (research) cycles link; those that need machinery outside these paths (MatMatMult, index sets /
 * sub-vectors, KSP monitors) abort with a clear message when called.
 *
 * Type constraints the reference imposes: PetscInt is a 32-bit int (int range[2] is
 * passed to VecGetOwnershipRange, src/solver.c:568,588), PetscScalar == PetscReal == double
 * (double* to VecGetArray, src/solver.c:1242,1255), PetscErrorCode is an int nobody checks.
 *
 * Execution model: one process, one GPU (MPI_Comm_size == 1).  The multi-GPU path of the product is the
 * slab-decomposed driver of include/mgsolve.h, not this shim.
 */
#ifndef MGPETSC_PETSCKSP_H
#define MGPETSC_PETSCKSP_H
#include <stdio.h>
#include <stdarg.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef int    PetscInt;
typedef double PetscScalar;
typedef double PetscReal;
typedef int    PetscErrorCode;
typedef int    PetscMPIInt;
typedef int    PetscLogStage;
typedef enum { PETSC_FALSE = 0, PETSC_TRUE = 1 } PetscBool;

typedef struct _p_Mat *Mat;
typedef struct _p_Vec *Vec;
typedef struct _p_KSP *KSP;
typedef struct _p_PC  *PC;
typedef struct _p_IS  *IS;
typedef struct _p_PetscViewer *PetscViewer;
typedef const char *KSPType;
typedef const char *PCType;
typedef const char *MatType;

#define PETSC_DEFAULT   (-2)
#define PETSC_DECIDE    (-1)
#define PETSC_DETERMINE PETSC_DECIDE
#define PETSC_NULL      NULL
#define PETSC_STDOUT    stdout

/* ---- MPI subset (src/solver.c:1273-1299,1526; 73 sites of MPI_Comm_size/rank) ---- */
typedef int MPI_Comm;
typedef int MPI_Datatype;
typedef struct { int MPI_SOURCE, MPI_TAG, MPI_ERROR; } MPI_Status;
#define MPI_COMM_WORLD   ((MPI_Comm)1)
#define PETSC_COMM_WORLD MPI_COMM_WORLD
#define PETSC_COMM_SELF  ((MPI_Comm)2)
#define MPI_DOUBLE       ((MPI_Datatype)8)
#define MPI_INT          ((MPI_Datatype)4)
#define MPI_STATUS_IGNORE ((MPI_Status *)0)
int    MPI_Comm_size(MPI_Comm comm, int *size);
int    MPI_Comm_rank(MPI_Comm comm, int *rank);
double MPI_Wtime(void);
int    MPI_Send(const void *buf, int count, MPI_Datatype type, int dest, int tag, MPI_Comm comm);
int    MPI_Recv(void *buf, int count, MPI_Datatype type, int source, int tag, MPI_Comm comm, MPI_Status *status);

/* ---- enums ---- */
typedef enum { NOT_SET_VALUES, INSERT_VALUES, ADD_VALUES } InsertMode;
typedef enum { MAT_FLUSH_ASSEMBLY = 1, MAT_FINAL_ASSEMBLY = 0 } MatAssemblyType;
typedef enum { NORM_1 = 0, NORM_2 = 1, NORM_FROBENIUS = 2, NORM_INFINITY = 3 } NormType;
typedef enum { KSP_NORM_DEFAULT = -1, KSP_NORM_NONE = 0, KSP_NORM_PRECONDITIONED = 1,
               KSP_NORM_UNPRECONDITIONED = 2, KSP_NORM_NATURAL = 3 } KSPNormType;
typedef enum { MAT_INITIAL_MATRIX, MAT_REUSE_MATRIX, MAT_IGNORE_MATRIX } MatReuse;
typedef enum { PETSC_COPY_VALUES, PETSC_OWN_POINTER, PETSC_USE_POINTER } PetscCopyMode;

#define KSPRICHARDSON "richardson"
#define KSPCHEBYSHEV  "chebyshev"
#define KSPGMRES      "gmres"
#define KSPCG         "cg"
#define KSPPREONLY    "preonly"
#define PCJACOBI      "jacobi"
#define PCNONE        "none"
#define PCILU         "ilu"
#define PCASM         "asm"
#define PCMG          "mg"
#define PCLU          "lu"

/* ---- viewers (content-free: the reference only passes them through) ---- */
PetscViewer PETSC_VIEWER_STDOUT_(MPI_Comm comm);
PetscViewer PETSC_VIEWER_DRAW_(MPI_Comm comm);
#define PETSC_VIEWER_STDOUT_WORLD PETSC_VIEWER_STDOUT_(PETSC_COMM_WORLD)
#define PETSC_VIEWER_STDOUT_SELF  PETSC_VIEWER_STDOUT_(PETSC_COMM_SELF)
#define PETSC_VIEWER_DRAW_WORLD   PETSC_VIEWER_DRAW_(PETSC_COMM_WORLD)

/* ---- system: src/poisson.c:29-59,135 ---- */
PetscErrorCode PetscInitialize(int *argc, char ***argv, const char file[], const char help[]);
PetscErrorCode PetscFinalize(void);
PetscErrorCode PetscOptionsGetInt(void *options, const char pre[], const char name[], PetscInt *ivalue, PetscBool *set);
PetscErrorCode PetscOptionsGetIntArray(void *options, const char pre[], const char name[], PetscInt ivalue[], PetscInt *nmax, PetscBool *set);
PetscErrorCode PetscOptionsGetReal(void *options, const char pre[], const char name[], PetscReal *dvalue, PetscBool *set);
PetscErrorCode PetscOptionsSetValue(void *options, const char name[], const char value[]);
PetscErrorCode PetscPrintf(MPI_Comm comm, const char format[], ...);
PetscErrorCode PetscSynchronizedPrintf(MPI_Comm comm, const char format[], ...);
PetscErrorCode PetscSynchronizedFlush(MPI_Comm comm, FILE *fd);
PetscErrorCode PetscLogStageRegister(const char name[], PetscLogStage *stage);     /* src/solver.c:1528 */
PetscErrorCode PetscLogStagePush(PetscLogStage stage);                             /* :1529 */
PetscErrorCode PetscLogStagePop(void);                                             /* :1551 */
PetscErrorCode PetscObjectSetOptionsPrefix(void *obj, const char prefix[]);

/* ---- Vec: src/solver.c:588-617,1255-1313,1459-1461,1512-1518 ---- */
PetscErrorCode VecCreateSeq(MPI_Comm comm, PetscInt n, Vec *v);
PetscErrorCode VecDuplicate(Vec v, Vec *newv);
PetscErrorCode VecDestroy(Vec *v);
PetscErrorCode VecGetSize(Vec v, PetscInt *n);
PetscErrorCode VecGetLocalSize(Vec v, PetscInt *n);
PetscErrorCode VecGetOwnershipRange(Vec v, PetscInt *low, PetscInt *high);
PetscErrorCode VecGetOwnershipRanges(Vec v, const PetscInt *ranges[]);
PetscErrorCode VecSetValue(Vec v, PetscInt row, PetscScalar value, InsertMode mode);
PetscErrorCode VecAssemblyBegin(Vec v);
PetscErrorCode VecAssemblyEnd(Vec v);
PetscErrorCode VecSet(Vec v, PetscScalar alpha);
PetscErrorCode VecCopy(Vec x, Vec y);
PetscErrorCode VecScale(Vec v, PetscScalar alpha);
PetscErrorCode VecAXPY(Vec y, PetscScalar alpha, Vec x);
PetscErrorCode VecAYPX(Vec y, PetscScalar alpha, Vec x);
PetscErrorCode VecAXPBYPCZ(Vec z, PetscScalar alpha, PetscScalar beta, PetscScalar gamma, Vec x, Vec y);
PetscErrorCode VecDot(Vec x, Vec y, PetscScalar *val);
PetscErrorCode VecTDot(Vec x, Vec y, PetscScalar *val);
PetscErrorCode VecNorm(Vec x, NormType type, PetscReal *val);
PetscErrorCode VecGetArray(Vec v, PetscScalar **a);
PetscErrorCode VecRestoreArray(Vec v, PetscScalar **a);
PetscErrorCode VecView(Vec v, PetscViewer viewer);
PetscErrorCode VecGetSubVector(Vec v, IS is, Vec *sub);
PetscErrorCode VecRestoreSubVector(Vec v, IS is, Vec *sub);

/* ---- IS (delayed cycles only; out of the hot path) ---- */
PetscErrorCode ISCreateGeneral(MPI_Comm comm, PetscInt n, const PetscInt idx[], PetscCopyMode mode, IS *is);
PetscErrorCode ISDestroy(IS *is);
PetscErrorCode ISView(IS is, PetscViewer viewer);

/* ---- Mat: src/solver.c:502-509,1071-1092,1131-1152,1172,1516,1535,1540 ---- */
PetscErrorCode MatCreateAIJ(MPI_Comm comm, PetscInt m, PetscInt n, PetscInt M, PetscInt N, PetscInt d_nz,
                            const PetscInt d_nnz[], PetscInt o_nz, const PetscInt o_nnz[], Mat *A);
PetscErrorCode MatSetValue(Mat A, PetscInt row, PetscInt col, PetscScalar value, InsertMode mode);
PetscErrorCode MatAssemblyBegin(Mat A, MatAssemblyType type);
PetscErrorCode MatAssemblyEnd(Mat A, MatAssemblyType type);
PetscErrorCode MatCreateVecs(Mat A, Vec *right, Vec *left);
PetscErrorCode MatGetSize(Mat A, PetscInt *m, PetscInt *n);
PetscErrorCode MatMult(Mat A, Vec x, Vec y);
PetscErrorCode MatMultAdd(Mat A, Vec x, Vec y, Vec z);
PetscErrorCode MatResidual(Mat A, Vec b, Vec x, Vec r);
PetscErrorCode MatScale(Mat A, PetscScalar a);
PetscErrorCode MatMatMult(Mat A, Mat B, MatReuse scall, PetscReal fill, Mat *C);
PetscErrorCode MatView(Mat A, PetscViewer viewer);
PetscErrorCode MatDestroy(Mat *A);

/* ---- KSP / PC: src/solver.c:1463-1510,1531-1546,1560-1570 ---- */
PetscErrorCode KSPCreate(MPI_Comm comm, KSP *ksp);
PetscErrorCode KSPSetType(KSP ksp, KSPType type);
PetscErrorCode KSPSetOperators(KSP ksp, Mat Amat, Mat Pmat);
PetscErrorCode KSPSetNormType(KSP ksp, KSPNormType normtype);
PetscErrorCode KSPSetTolerances(KSP ksp, PetscReal rtol, PetscReal abstol, PetscReal dtol, PetscInt maxits);
PetscErrorCode KSPSetFromOptions(KSP ksp);
PetscErrorCode KSPSetInitialGuessNonzero(KSP ksp, PetscBool flg);
PetscErrorCode KSPSolve(KSP ksp, Vec b, Vec x);
PetscErrorCode KSPBuildResidual(KSP ksp, Vec t, Vec v, Vec *V);
PetscErrorCode KSPView(KSP ksp, PetscViewer viewer);
PetscErrorCode KSPDestroy(KSP *ksp);
PetscErrorCode KSPGetPC(KSP ksp, PC *pc);
PetscErrorCode KSPGetIterationNumber(KSP ksp, PetscInt *its);
PetscErrorCode KSPSetResidualHistory(KSP ksp, PetscReal a[], PetscInt na, PetscBool reset);
PetscErrorCode KSPMonitorSet(KSP ksp, PetscErrorCode (*monitor)(KSP, PetscInt, PetscReal, void *), void *mctx,
                             PetscErrorCode (*monitordestroy)(void **));
PetscErrorCode KSPRichardsonSetScale(KSP ksp, PetscReal scale);
PetscErrorCode KSPChebyshevSetEigenvalues(KSP ksp, PetscReal emax, PetscReal emin);
PetscErrorCode PCSetType(PC pc, PCType type);
PetscErrorCode PCMGSetLevels(PC pc, PetscInt levels, MPI_Comm *comms);
PetscErrorCode PCMGGetCoarseSolve(PC pc, KSP *ksp);
PetscErrorCode PCMGGetSmoother(PC pc, PetscInt l, KSP *ksp);
PetscErrorCode PCMGSetInterpolation(PC pc, PetscInt l, Mat mat);
PetscErrorCode PCMGSetRestriction(PC pc, PetscInt l, Mat mat);
PetscErrorCode PCMGSetR(PC pc, PetscInt l, Vec c);
PetscErrorCode PCMGSetRhs(PC pc, PetscInt l, Vec c);
PetscErrorCode PCMGSetX(PC pc, PetscInt l, Vec c);

#ifdef __cplusplus
}
#endif
#endif
