// Port of the reference's test/test_mpc_12state.cpp with assertions, against the C++ look-alike of
// `class MPC12STATE` (include/tg_mpc_classes.hpp) over libtg_mpc.so: N = 10, dt = 0.1, weights
// 1000 / 10 (yaw input 1), limits 12-5-5 / 5-5-10 / 10-10-10, x0 = (0.03, -0.04, 0.1), reference
// x = 4, y = 0.5, z = 40, three chained solves, roll-out check (test_mpc_12state.cpp:186-213) and
// the CSV dump.  Exit code 0 = all checks passed; needs a GPU.
#include <cmath>
#include <cstdio>
#include <fstream>
#include <string>

#include "tg_mpc_classes.hpp"

#define CHECK(cond)                                                                  \
    do {                                                                             \
        if (!(cond)) { std::printf("CHECK FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); return 1; } \
    } while (0)

int main()
{
    tg::MPC12STATE *mpc = new tg::MPC12STATE();
    const int mpcWindow = 10;
    CHECK(mpc->setDt(0.1) && mpc->setMPCWindow(mpcWindow));
    CHECK(mpc->setXYStateWeight(1000.0) && mpc->setZStateWeight(1000.0) && mpc->setYawStateWeight(1000.0));
    CHECK(mpc->setXYInputWeight(10.0) && mpc->setZInputWeight(10.0) && mpc->setYawInputWeight(1.0));
    CHECK(!mpc->initMPCProblem());                                  // limits missing
    CHECK(mpc->setXYMaxVel(12.0) && mpc->setZMaxVel(5.0) && mpc->setYawMaxVel(5.0));
    CHECK(mpc->setXYMaxAccel(5.0) && mpc->setZMaxAccel(5.0) && mpc->setYawMaxAccel(10.0));
    CHECK(mpc->setXYMaxJerk(10.0) && mpc->setZMaxJerk(10.0) && mpc->setYawMaxJerk(10.0));
    CHECK(mpc->initMPCProblem());

    const int NS = tg::MPC12STATE::NUM_OF_STATES, NI = tg::MPC12STATE::NUM_OF_INPUTS;
    CHECK(!mpc->mpcLoop());                                         // no state / reference yet
    tg::VectorXd current_state(NS);
    current_state(0) = 0.03; current_state(3) = -0.04; current_state(6) = 0.1;
    CHECK(mpc->setCurrentState(current_state));
    CHECK(!mpc->mpcLoop());                                         // no reference yet
    tg::MatrixXd ref_traj(NS * (mpcWindow + 1), 1);
    for (int i = 0; i < mpcWindow + 1; i++) {
        ref_traj(i * NS + 6, 0) = 40.0; ref_traj(i * NS + 0, 0) = 4.0; ref_traj(i * NS + 3, 0) = 0.5;
    }
    CHECK(mpc->setReferenceTraj(ref_traj));
    CHECK(mpc->mpcLoop());
    std::printf("iterations z / xy / yaw: %d %d %d\n", mpc->last_iters[0], mpc->last_iters[1], mpc->last_iters[2]);

    tg::VectorXd X = mpc->getOptimalStateTraj(), U = mpc->getOptimalControlTraj();
    CHECK(X.size() == NS * (mpcWindow + 1) && U.size() == NI * mpcWindow);
    tg::MatrixXd A = mpc->getTransitionMatrix(), B = mpc->getInputMatrix();
    CHECK(A.rows() == 12 && A.cols() == 12 && B.rows() == 12 && B.cols() == 4);
    tg::VectorXd x0 = current_state;
    for (int i = 0; i < mpcWindow; i++) x0 = A * x0 + B * U.segment(i * NI, NI);
    const double err = (X.segment(NS * mpcWindow, NS) - x0).norm();
    std::printf("At step = %d, error between simulated and optimal final state = %g\n", mpcWindow, err);
    CHECK(err < 5e-3);
    for (int i = 0; i <= mpcWindow; i++) {
        CHECK(std::fabs(X(i * NS + 7)) <= 5.0 + 5e-2 && std::fabs(X(i * NS + 8)) <= 5.0 + 5e-2);   // eps_abs + eps_rel |z|, z ~ 40
    }
    for (int i = 0; i < U.size(); i++) CHECK(std::fabs(U(i)) <= 10.0 + 1e-3);
    CHECK(X(NS * mpcWindow + 6) > 0.5);                             // climbing towards z = 40
    CHECK(mpc->getR().rows() == 4 && mpc->getQ().rows() == 12);
    CHECK(mpc->getXYContraintsMatrix().rows() == 20 * mpcWindow + 12 && mpc->getZHessianMatrix().rows() == 4 * mpcWindow + 3);
    CHECK(mpc->getZLowerBounds()(0) == -0.1 && mpc->getXYGradient()(0) == -4000.0);

    const std::string path = "/tmp/test_mpc_12state.csv";
    mpc->setOutputFilePath(path);
    mpc->saveMPCSolutionsToFile();
    std::ifstream f(path);
    std::string line;
    int lines = 0, commas_hdr = 0, commas_row1 = 0, commas_row2 = 0;
    while (std::getline(f, line)) {
        int c = 0;
        for (char ch : line) c += ch == ',';
        if (lines == 0) commas_hdr = c;
        if (lines == 1) commas_row1 = c;
        if (lines == 2) commas_row2 = c;
        lines++;
    }
    CHECK(lines == mpcWindow + 2 && commas_hdr == 46 && commas_row1 == 28 && commas_row2 == 46);
    delete mpc;
    std::printf("test_mpc_12state: OK\n");
    return 0;
}
