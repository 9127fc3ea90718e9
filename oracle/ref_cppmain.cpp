// ref_cppmain.cpp -- builds the REAL reference CPU path (C++/main.cpp) into
// oracle/_ref/libref_cppmain.so so tests can call it.  TEST INFRASTRUCTURE ONLY.
//
// The reference source is compiled where it lies under /root/reference (it is
// #included by absolute path via -DMPQR_REF_MAIN_CPP=...; nothing is copied into
// this repo).  Its own `main` is renamed so the two functions it defines,
//   VectorXd householder(VectorXd u)                       C++/main.cpp:5-14
//   void qr_factorization(MatrixXd& A, MatrixXd& Q)        C++/main.cpp:16-43
// become callable through the extern "C" wrappers below.  Eigen 3.4.0 is the
// copy vendored by the reference under Cuda/QR/Solver/Eigen (-I on the command
// line in oracle/Makefile).
#define main mpqr_reference_main_unused
#include MPQR_REF_MAIN_CPP
#undef main

extern "C" {

// A, Q: column-major n x n doubles (Eigen's MatrixXd layout).  Q must hold the
// identity on entry, exactly as C++/main.cpp:58 does.
void ref_qr_factorization(double* A, double* Q, int n) {
    Eigen::MatrixXd Am = Eigen::Map<Eigen::MatrixXd>(A, n, n);
    Eigen::MatrixXd Qm = Eigen::Map<Eigen::MatrixXd>(Q, n, n);
    qr_factorization(Am, Qm);
    Eigen::Map<Eigen::MatrixXd>(A, n, n) = Am;
    Eigen::Map<Eigen::MatrixXd>(Q, n, n) = Qm;
}

void ref_householder(const double* u, int len, double* w) {
    Eigen::VectorXd uv = Eigen::Map<const Eigen::VectorXd>(u, len);
    Eigen::VectorXd wv = householder(uv);
    for (int i = 0; i < len; i++) w[i] = wv[i];
}

// the reference program itself (prints ||A-QR||/||A|| of its hard-coded 3x3)
int ref_main() { return mpqr_reference_main_unused(); }

}
