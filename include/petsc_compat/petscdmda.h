/* petscdmda.h (compat): everything lives in petsc.h */
#include <petsc.h>
