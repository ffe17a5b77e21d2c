// Port of the reference's test/test_mpc_6dof.cpp with assertions, against the C++ look-alike of
// `class MPC` (include/tg_mpc_classes.hpp) over libtg_mpc.so.  Same sequence as the reference's
// main: configure, initMPCProblem, current state, reference (x = 0.1, z = 1), updateMPC, mpcLoop,
// roll x+ = A x + B u with the returned controls and compare the final state
// (test_mpc_6dof.cpp:115-135).  Exit code 0 = all checks passed; needs a GPU.
#include <cmath>
#include <cstdio>
#include <cstdlib>

#include "tg_mpc_classes.hpp"

#define CHECK(cond)                                                                  \
    do {                                                                             \
        if (!(cond)) { std::printf("CHECK FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); return 1; } \
    } while (0)

int main()
{
    tg::MPC *mpc = new tg::MPC();
    mpc->setDebug(false);
    CHECK(mpc->setDt(0.1));
    const int mpcWindow = 20;
    CHECK(mpc->setMPCWindow(mpcWindow));
    CHECK(mpc->setStateWeight(7000.0));
    CHECK(mpc->setInputWeight(500.0));
    CHECK(!mpc->setStateWeight(-1.0));                        // rejected, old value kept
    std::vector<double> maxVel = {10.0, 10.0, 10.0}, maxAccel = {5.0, 5.0, 5.0};
    CHECK(mpc->setMaxVel(maxVel) && mpc->setMaxAccel(maxAccel));
    CHECK(!mpc->setMaxVel(std::vector<double>{1.0, 2.0}));
    CHECK(mpc->setMinimumAltitude(-1.0));
    CHECK(!mpc->mpcLoop());                                   // not initialised yet
    CHECK(mpc->initMPCProblem());

    tg::VectorXd current_state(6);
    current_state(0) = 0.1;
    CHECK(mpc->setCurrentState(current_state));
    const int NS = tg::MPC::NUM_OF_STATES, NI = tg::MPC::NUM_OF_INPUTS;
    tg::MatrixXd ref_traj(NS * (mpcWindow + 1), 1);
    for (int i = 0; i < mpcWindow + 1; i++) { ref_traj(i * NS + 2, 0) = 1.0; ref_traj(i * NS + 0, 0) = 0.1; }
    CHECK(mpc->setReferenceTraj(ref_traj));
    tg::MatrixXd wrong(NS * mpcWindow, 1);
    CHECK(!mpc->setReferenceTraj(wrong));
    CHECK(mpc->updateMPC());
    CHECK(mpc->mpcLoop());

    tg::VectorXd X = mpc->getOptimalStateTraj(), U = mpc->getOptimalControlTraj();
    CHECK(X.size() == NS * (mpcWindow + 1) && U.size() == NI * mpcWindow);
    tg::VectorXd x0 = current_state;
    for (int i = 0; i < mpcWindow; i++)
        x0 = mpc->getTransitionMatrix() * x0 + mpc->getInputMatrix() * U.segment(i * NI, NI);
    tg::VectorXd xN = X.segment(NS * mpcWindow, NS);
    const double err = (xN - x0).norm();
    std::printf("At step = %d, error between simulated and optimal final state = %g (iters %d)\n", mpcWindow, err,
                mpc->last_iters);
    CHECK(err < 5e-3);
    for (int i = 0; i < U.size(); i++) CHECK(std::fabs(U(i)) <= 5.0 + 1e-6);
    for (int i = 0; i <= mpcWindow; i++)
        for (int k = 3; k < 6; k++) CHECK(std::fabs(X(i * NS + k)) <= 10.0 + 1e-6);
    // the vehicle climbs towards z = 1 and stays at x = 0.1
    CHECK(X(NS * mpcWindow + 2) > 0.5 && std::fabs(X(NS * mpcWindow + 0) - 0.1) < 0.05);
    // problem data getters
    CHECK(mpc->getHessianMatrix().rows() == 186 && mpc->getContraintsMatrix().rows() == 312);
    CHECK(mpc->getGradient()(2) == -7000.0);
    CHECK(mpc->getLowerBounds()(0) == -0.1 && mpc->getUpperBounds()(0) == -0.1);
    CHECK(mpc->getQ()(0, 0) == 7000.0 && mpc->getR()(2, 2) == 500.0 && mpc->getQ()(3, 3) == 0.0);
    // second call continues from the previous iterates (setWarmStart(true), mpc_6dof.cpp:528)
    CHECK(mpc->mpcLoop() && mpc->last_iters <= 25);
    mpc->setOutputFilePath("/tmp/test_mpc_6dof_data.txt");
    mpc->saveMPCDataToFile();
    delete mpc;
    std::printf("test_mpc_6dof: OK\n");
    return 0;
}
