// Header-only C++ look-alikes of the reference's `class MPC`
// (include/trajectory_generation/mpc_6dof.hpp:91-404) and `class MPC12STATE`
// (include/trajectory_generation/mpc_12state.hpp:492-683) on top of the C ABI in tg_mpc.h:
// same method names, argument meaning, bool returns and call-order contract, so mpc_ros.cpp /
// mpc_12state_ros.cpp and the two test mains compile against them unchanged.  Every solve is one
// B = 1 call into libtg_mpc.so (MI355X); there is no CPU solver behind these classes.
//
// Vector / matrix types: Eigen's when <Eigen/Dense> is on the include path (as in the
// reference's build), otherwise the minimal tg::VectorXd / tg::MatrixXd below, which cover what
// the reference's callers use (size, (), segment, block-free products, norm, <<).
#pragma once
#include <cmath>
#include <cstdio>
#include <fstream>
#include <iostream>
#include <string>
#include <vector>

#include "tg_mpc.h"

#if defined(TG_USE_EIGEN) || (defined(__has_include) && __has_include(<Eigen/Dense>) && !defined(TG_NO_EIGEN))
#include <Eigen/Dense>
namespace tg {
using VectorXd = Eigen::VectorXd;
using MatrixXd = Eigen::MatrixXd;
inline VectorXd make_vec(const double *p, int n) { return Eigen::Map<const Eigen::VectorXd>(p, n); }
inline MatrixXd make_mat(const double *rowmajor, int r, int c)
{
    return Eigen::Map<const Eigen::Matrix<double, Eigen::Dynamic, Eigen::Dynamic, Eigen::RowMajor>>(rowmajor, r, c);
}
template <class V> inline std::vector<double> to_std(const V &v)
{
    std::vector<double> o((size_t)v.size());
    for (long i = 0; i < (long)v.size(); i++) o[(size_t)i] = v(i);
    return o;
}
} // namespace tg
#else
namespace tg {
class VectorXd {
public:
    VectorXd() {}
    explicit VectorXd(int n) : v_((size_t)n, 0.0) {}
    VectorXd(std::initializer_list<double> l) : v_(l) {}
    int size() const { return (int)v_.size(); }
    int rows() const { return size(); }
    int cols() const { return 1; }
    void resize(int n) { v_.assign((size_t)n, 0.0); }
    void resize(int n, int) { resize(n); }
    void setZero() { std::fill(v_.begin(), v_.end(), 0.0); }
    double &operator()(int i) { return v_[(size_t)i]; }
    double operator()(int i) const { return v_[(size_t)i]; }
    double &operator()(int i, int) { return v_[(size_t)i]; }
    double operator()(int i, int) const { return v_[(size_t)i]; }
    double *data() { return v_.data(); }
    const double *data() const { return v_.data(); }
    VectorXd segment(int first, int n) const
    {
        VectorXd o(n);
        for (int i = 0; i < n; i++) o(i) = v_[(size_t)(first + i)];
        return o;
    }
    double norm() const
    {
        double s = 0;
        for (double x : v_) s += x * x;
        return std::sqrt(s);
    }
    VectorXd operator+(const VectorXd &b) const { VectorXd o(size()); for (int i = 0; i < size(); i++) o(i) = v_[(size_t)i] + b(i); return o; }
    VectorXd operator-(const VectorXd &b) const { VectorXd o(size()); for (int i = 0; i < size(); i++) o(i) = v_[(size_t)i] - b(i); return o; }
private:
    std::vector<double> v_;
};
class MatrixXd {
public:
    MatrixXd() {}
    MatrixXd(int r, int c) : r_(r), c_(c), a_((size_t)r * c, 0.0) {}
    int rows() const { return r_; }
    int cols() const { return c_; }
    int size() const { return r_ * c_; }
    void resize(int r, int c) { r_ = r; c_ = c; a_.assign((size_t)r * c, 0.0); }
    void setZero() { std::fill(a_.begin(), a_.end(), 0.0); }
    double &operator()(int i, int j) { return a_[(size_t)i * c_ + j]; }
    double operator()(int i, int j) const { return a_[(size_t)i * c_ + j]; }
    VectorXd operator*(const VectorXd &x) const
    {
        VectorXd o(r_);
        for (int i = 0; i < r_; i++) {
            double s = 0;
            for (int j = 0; j < c_; j++) s += (*this)(i, j) * x(j);
            o(i) = s;
        }
        return o;
    }
private:
    int r_ = 0, c_ = 0;
    std::vector<double> a_;   // row-major
};
inline std::ostream &operator<<(std::ostream &os, const VectorXd &v)
{
    for (int i = 0; i < v.size(); i++) os << v(i) << (i + 1 < v.size() ? "\n" : "");
    return os;
}
inline std::ostream &operator<<(std::ostream &os, const MatrixXd &m)
{
    for (int i = 0; i < m.rows(); i++) {
        for (int j = 0; j < m.cols(); j++) os << m(i, j) << (j + 1 < m.cols() ? " " : "");
        if (i + 1 < m.rows()) os << "\n";
    }
    return os;
}
inline VectorXd make_vec(const double *p, int n) { VectorXd v(n); for (int i = 0; i < n; i++) v(i) = p[i]; return v; }
inline MatrixXd make_mat(const double *rowmajor, int r, int c)
{
    MatrixXd m(r, c);
    for (int i = 0; i < r; i++) for (int j = 0; j < c; j++) m(i, j) = rowmajor[(size_t)i * c + j];
    return m;
}
template <class V> inline std::vector<double> to_std(const V &v)
{
    std::vector<double> o((size_t)v.size());
    for (int i = 0; i < (int)v.size(); i++) o[(size_t)i] = v(i);
    return o;
}
} // namespace tg
#endif

namespace tg {

inline void print_error(const std::string &m) { std::printf("\033[1;31m[ERROR] %s\033[0m\n", m.c_str()); }
inline void print_warn(const std::string &m) { std::printf("\033[1;33m[INFO] %s\033[0m\n", m.c_str()); }
inline void print_info(const std::string &m) { std::printf("\033[1;32m[INFO] %s\033[0m\n", m.c_str()); }

// flattens a vector or an n x 1 matrix
template <class M> inline std::vector<double> flatten(const M &m)
{
    std::vector<double> o((size_t)m.size());
    const int r = (int)m.rows();
    for (int i = 0; i < (int)m.size(); i++) o[(size_t)i] = m(i % r, i / r);
    return o;
}

// ------------------------------------------------------------------ class MPC (6-DOF)
class MPC {
public:
    static constexpr int NUM_OF_STATES = 6, NUM_OF_INPUTS = 3;   // mpc_6dof.hpp:77-78

    explicit MPC(int device = 0) : _device(device) {}             // defaults: mpc_6dof.cpp:36-52
    ~MPC() { tg_mpc6_destroy(_h); }
    MPC(const MPC &) = delete;
    MPC &operator=(const MPC &) = delete;

    bool setDt(double dt)
    {
        if (dt > 0) { _dt = dt; return true; }
        print_error("[MPC::setDt] dt < 0. Defaulting to 0.1");
        _dt = 0.1;
        return false;
    }
    void setDebug(bool d) { _debug = d; }
    bool setMPCWindow(int N)
    {
        if (N > 0) { _mpcWindow = N; return true; }
        print_error("_mpcWindow < 0. Default to 10");
        _mpcWindow = 10;
        return false;
    }
    bool setStateWeight(double w) { return weight(w, _state_weight, "state_weight"); }
    bool setInputWeight(double w) { return weight(w, _input_weight, "_input_weight"); }
    bool setSmoothInputWeight(double w) { return weight(w, _smooth_input_weight, "_smooth_input_weight"); }
    void enableControlSmoothing(bool b) { _enable_control_smoothing = b; }
    bool setAltAboveTarget(double h) { return weight(h, _alt_above_target, "_alt_above_target"); }
    bool setMinimumAltitude(double h) { _minAltitude = h; return true; }
    bool setMaxVel(const std::vector<double> &v)
    {
        if (v.size() != 3) { print_error("maxVel vector size != 3"); return false; }
        _maxVel = v;
        return true;
    }
    bool setMaxAccel(const std::vector<double> &a)
    {
        if (a.size() != 3) { print_error("maxAccel vector size != 3"); return false; }
        _maxAccel = a;
        return true;
    }
    template <class V> bool setCurrentState(const V &x)
    {
        if ((int)x.size() != NUM_OF_STATES) { print_error("State vector size != 6"); return false; }
        _current_state = flatten(x);
        _state_received = true;
        return true;
    }
    template <class V> bool setCurrentAccel(const V &a)
    {
        if ((int)a.size() != 3) { print_error("Input acceleration vector size != 3"); return false; }
        _current_accel = flatten(a);
        return true;
    }
    // the argument's size is checked (the reference checks its own member, mpc_6dof.cpp:796)
    template <class M> bool setReferenceTraj(const M &v)
    {
        if ((int)v.size() != NUM_OF_STATES * (_mpcWindow + 1)) {
            print_error("[MPC::set_referenceTraj] input matrix size is wrong");
            return false;
        }
        _referenceTraj = flatten(v);
        return true;
    }
    void setOutputFilePath(std::string path) { _outputCSVFile = path; }

    // mpc_6dof.cpp:552-580: horizon, dt, weights and limits are frozen from here on
    bool initMPCProblem()
    {
        if (_maxVel.size() != 3 || _maxAccel.size() != 3) {
            print_error("[MPC::initMPCProblem] setMaxVel / setMaxAccel must be called first");
            return false;
        }
        tg_mpc6_params p{};
        p.N = _mpcWindow; p.dt = _dt; p.state_weight = _state_weight; p.input_weight = _input_weight;
        p.smooth_input_weight = _smooth_input_weight; p.enable_control_smoothing = _enable_control_smoothing ? 1 : 0;
        p.min_altitude = _minAltitude;
        for (int a = 0; a < 3; a++) { p.max_vel[a] = _maxVel[(size_t)a]; p.max_accel[a] = _maxAccel[(size_t)a]; }
        tg_mpc6_destroy(_h); _h = nullptr;
        if (tg_mpc6_create(&p, _opts_set ? &_opts : nullptr, _device, &_h) != TG_OK) {
            print_error(std::string("[MPC::initMPCProblem] MPC initialization is not successful. ") + tg_last_error());
            return false;
        }
        tg_mpc6_dims(_h, &_n, &_m);
        _current_state.assign(NUM_OF_STATES, 0.0);
        _referenceTraj.assign((size_t)NUM_OF_STATES * (_mpcWindow + 1), 0.0);
        _optimal_state_traj.assign((size_t)NUM_OF_STATES * (_mpcWindow + 1), 0.0);
        _optimal_control_traj.assign((size_t)NUM_OF_INPUTS * _mpcWindow, 0.0);
        _is_MPC_initialized = refresh();
        return _is_MPC_initialized;
    }
    bool updateMPC() { return _is_MPC_initialized && refresh(); }   // updateQP, mpc_6dof.cpp:584
    // mpc_6dof.cpp:616-649; warm start as the reference (:528)
    bool mpcLoop()
    {
        if (!_is_MPC_initialized) { print_warn("[MPC::mpcLoop] MPC controller is not initialized. Skipping MPC loop."); return false; }
        if (!_state_received) { print_warn("[MPC::mpcLoop] Drone state is not received yet. Skipping MPC loop."); return false; }
        if (!refresh()) return false;
        std::vector<double> X(_optimal_state_traj.size()), U(_optimal_control_traj.size());
        int rc = tg_mpc6_solve_host(_h, 1, _current_state.data(), _referenceTraj.data(), X.data(), U.data(),
                                    &last_status, &last_iters, 1);
        if (rc != TG_OK || last_status != TG_SOLVED) {
            print_error("[MPC::mpcLoop] MPC solution is not found");
            return false;                                           // previous trajectories stay
        }
        _optimal_state_traj.swap(X);
        _optimal_control_traj.swap(U);
        return true;
    }

    VectorXd getOptimalStateTraj() { return make_vec(_optimal_state_traj.data(), (int)_optimal_state_traj.size()); }
    VectorXd getOptimalControlTraj() { return make_vec(_optimal_control_traj.data(), (int)_optimal_control_traj.size()); }
    int getNumOfStates() { return NUM_OF_STATES; }
    int getNumOfInputs() { return NUM_OF_INPUTS; }
    std::vector<double> getMaxAccel() { return _maxAccel; }
    std::vector<double> getMaxVel() { return _maxVel; }
    MatrixXd getTransitionMatrix() { return mat(TG_MAT_AD, 6, 6); }
    MatrixXd getInputMatrix() { return mat(TG_MAT_BD, 6, 3); }
    MatrixXd getQ() { return mat(TG_MAT_Q, 6, 6); }
    MatrixXd getR() { return mat(TG_MAT_R, 3, 3); }
    MatrixXd getHessianMatrix() { return mat(TG_MAT_P, _n, _n); }
    MatrixXd getContraintsMatrix() { return mat(TG_MAT_A, _m, _n); }
    VectorXd getGradient() { return make_vec(_gradient.data(), (int)_gradient.size()); }
    VectorXd getLowerBounds() { return make_vec(_lowerBounds.data(), (int)_lowerBounds.size()); }
    VectorXd getUpperBounds() { return make_vec(_upperBounds.data(), (int)_upperBounds.size()); }

    // mpc_6dof.cpp:739-791
    void saveMPCDataToFile()
    {
        std::ofstream f(_outputCSVFile);
        if (!f.is_open()) { print_error("[MPCROS::saveMPCDataToFile] Could not open file " + _outputCSVFile); return; }
        const char *sep = "\n------------------------------------------\n";
        f << "Initial state, x(0): \n" << make_vec(_current_state.data(), 6) << sep;
        f << " A : \n" << getTransitionMatrix() << sep << "B : \n" << getInputMatrix() << sep;
        f << "Q : \n" << getQ() << sep << "R : \n" << getR() << sep;
        f << "Hessian matrix, P: \n" << getHessianMatrix() << sep;
        f << "Constarints matrix, Ac: \n" << getContraintsMatrix() << sep;
        f << "Lower bounds, l: \n" << getLowerBounds() << sep << "Upper bounds, l: \n" << getUpperBounds() << sep;
        f << "gradient, q: \n" << getGradient() << sep;
        f << "Optimal state trajectory, X: \n" << getOptimalStateTraj() << sep;
        f << "Optimal control trajectory, U: \n" << getOptimalControlTraj() << sep;
        print_info("[MPC::saveMPCDataToFile] Saved MPC solutions to file: " + _outputCSVFile);
    }

    // not in the reference: solver options of the engine, to be set before initMPCProblem
    void setSolverOptions(const tg_solver_opts &o) { _opts = o; _opts_set = true; }
    int last_status = TG_UNSOLVED, last_iters = 0;

private:
    static bool weight(double w, double &dst, const char *name)
    {
        if (w < 0.0) { print_error(std::string(name) + " < 0"); return false; }
        dst = w;
        return true;
    }
    bool refresh()   // castMPCToQPGradient / castMPCToQPConstraintBounds for the current x0 / ref
    {
        _gradient.assign((size_t)_n, 0.0); _lowerBounds.assign((size_t)_m, 0.0); _upperBounds.assign((size_t)_m, 0.0);
        return tg_mpc6_assemble_host(_h, 1, _current_state.data(), _referenceTraj.data(), _gradient.data(),
                                     _lowerBounds.data(), _upperBounds.data()) == TG_OK;
    }
    MatrixXd mat(int which, int r, int c)
    {
        std::vector<double> b((size_t)r * c, 0.0);
        if (_h) tg_mpc6_get(_h, which, b.data());
        return make_mat(b.data(), r, c);
    }
    bool _debug = true, _state_received = false, _enable_control_smoothing = true, _is_MPC_initialized = false;
    double _dt = 0.05, _alt_above_target = 1.0, _state_weight = 1.0, _input_weight = 0.1, _smooth_input_weight = 10.0,
           _minAltitude = 1.0;
    int _mpcWindow = 20, _device = 0, _n = 0, _m = 0;
    std::vector<double> _maxVel, _maxAccel, _current_state = std::vector<double>(6, 0.0), _current_accel, _referenceTraj,
                        _optimal_state_traj, _optimal_control_traj, _gradient, _lowerBounds, _upperBounds;
    std::string _outputCSVFile;
    tg_mpc6 *_h = nullptr;
    tg_solver_opts _opts{};
    bool _opts_set = false;
};

// ------------------------------------------------------------------ class MPC12STATE
class MPC12STATE {
public:
    static constexpr int NUM_OF_STATES = 12, NUM_OF_INPUTS = 4;  // mpc_12state.hpp:86-87

    explicit MPC12STATE(int device = 0) : _device(device) {}     // defaults: mpc_12state.cpp:41-64
    ~MPC12STATE() { tg_mpc12_destroy(_h); }
    MPC12STATE(const MPC12STATE &) = delete;
    MPC12STATE &operator=(const MPC12STATE &) = delete;

    bool setDt(double dt)
    {
        if (dt > 0) { _dt = dt; return true; }
        print_error("[MPC12STATE::setDt] dt < 0. Defaulting to 0.1");
        _dt = 0.1;
        return false;
    }
    void setDebug(bool d) { _debug = d; }
    bool setMPCWindow(int N)
    {
        if (N > 0) { _mpcWindow = N; return true; }
        print_error("_mpcWindow < 0. Default to 10");
        _mpcWindow = 10;
        return false;
    }
    bool setXYStateWeight(double w) { return weight(w, _p.xy_state_weight, "xy_state_weight"); }
    bool setZStateWeight(double w) { return weight(w, _p.z_state_weight, "z_state_weight"); }
    bool setYawStateWeight(double w) { return weight(w, _p.yaw_state_weight, "yaw_state_weight"); }
    bool setXYInputWeight(double w) { return weight(w, _p.xy_input_weight, "_xy_input_weight"); }
    bool setZInputWeight(double w) { return weight(w, _p.z_input_weight, "_z_input_weight"); }
    bool setYawInputWeight(double w) { return weight(w, _p.yaw_input_weight, "_yaw_input_weight"); }
    bool setXYSmoothInputWeight(double w) { return weight(w, _p.xy_smooth_input_weight, "_xy_smooth_input_weight"); }
    bool setZSmoothInputWeight(double w) { return weight(w, _p.z_smooth_input_weight, "_z_smooth_input_weight"); }
    bool setYawSmoothInputWeight(double w) { return weight(w, _p.yaw_smooth_input_weight, "_yaw_smooth_input_weight"); }
    void enableControlSmoothing(bool b) { _enable_control_smoothing = b; }
    bool setAltAboveTarget(double h)
    {
        if (h < 0) { print_error("_alt_above_target < 0"); return false; }
        _alt_above_target = h;
        return true;
    }
    bool setMinimumAltitude(double h) { _minAltitude = h; return true; }   // unused by every stage, as in the reference (:260-270)
    // out-of-range limits install the reference's fall-back constants (mpc_12state.cpp:1817-1932)
    bool setXYMaxVel(double v) { return limit(v, _p.xy_max_vel, 0, "maxXYVel", 10.0); }
    bool setZMaxVel(double v) { return limit(v, _p.z_max_vel, 1, "maxZVel", 10.0); }
    bool setYawMaxVel(double v) { return limit(v, _p.yaw_max_vel, 2, "maxYawVel", 100.0); }
    bool setXYMaxAccel(double v) { return limit(v, _p.xy_max_accel, 3, "maxXYAccel", 5.0); }
    bool setZMaxAccel(double v) { return limit(v, _p.z_max_accel, 4, "maxZAccel", 5.0); }
    bool setYawMaxAccel(double v) { return limit(v, _p.yaw_max_accel, 5, "maxYawAccel", 100.0); }
    bool setXYMaxJerk(double v) { return limit(v, _p.xy_max_jerk, 6, "_xy_MaxJerk", 5.0); }
    bool setZMaxJerk(double v) { return limit(v, _p.z_max_jerk, 7, "_z_MaxJerk", 5.0); }
    bool setYawMaxJerk(double v) { return limit(v, _p.yaw_max_jerk, 8, "_yaw_MaxJerk", 100.0); }
    template <class V> bool setCurrentState(const V &x)
    {
        if ((int)x.size() != NUM_OF_STATES) { print_error("State vector size != 12"); return false; }
        _current_state = flatten(x);
        _state_received = true;
        return true;
    }
    template <class M> bool setReferenceTraj(const M &v)
    {
        if ((int)v.size() != NUM_OF_STATES * (_mpcWindow + 1)) {
            print_error("[MPC12STATE::setReferenceTraj] input matrix size is wrong");
            return false;
        }
        _referenceTraj = flatten(v);
        _received_refTraj = true;
        return true;
    }
    void setOutputFilePath(std::string path) { _outputCSVFile = path; }

    // mpc_12state.cpp:1282-1350
    bool initMPCProblem()
    {
        if (_limits_set != 0x1ff) {
            print_error("[MPC12STATE::initMPCProblem] all velocity / acceleration / jerk limits must be set first");
            return false;
        }
        _p.N = _mpcWindow; _p.dt = _dt; _p.enable_control_smoothing = _enable_control_smoothing ? 1 : 0;
        tg_mpc12_destroy(_h); _h = nullptr;
        if (tg_mpc12_create(&_p, _opts_set ? &_opts : nullptr, nullptr, _device, &_h) != TG_OK) {
            print_error(std::string("[MPC12STATE::initMPCProblem] MPC initialization is not successful. ") + tg_last_error());
            return false;
        }
        _current_state.assign(NUM_OF_STATES, 0.0);
        _referenceTraj.assign((size_t)NUM_OF_STATES * (_mpcWindow + 1), 0.0);
        _x_opt.assign((size_t)NUM_OF_STATES * (_mpcWindow + 1), 0.0);
        _u_opt.assign((size_t)NUM_OF_INPUTS * _mpcWindow, 0.0);
        _received_refTraj = false; _state_received = false;        // initVariables, :1185-1186
        _is_MPC_initialized = true;
        return true;
    }
    // mpc_12state.cpp:1374-1453: Z, XY, Yaw; cold start as the reference (:1025,1045,1065)
    bool mpcLoop()
    {
        if (!_is_MPC_initialized) { print_warn("[MPC12STATE::mpcLoop] MPC controller is not initialized. Skipping MPC loop."); return false; }
        if (!_state_received) { print_warn("[MPC12STATE::mpcLoop] curent state is not received. Skipping MPC loop."); return false; }
        if (!_received_refTraj) { print_warn("[MPC12STATE::mpcLoop] Reference trajectory is not received. Skipping MPC loop."); return false; }
        std::vector<double> X(_x_opt.size()), U(_u_opt.size());
        int rc = tg_mpc12_solve_host(_h, 1, _current_state.data(), _referenceTraj.data(), X.data(), U.data(),
                                     last_status, last_iters, 0);
        static const char *names[3] = {"Z", "XY", "Yaw"};
        for (int k = 0; k < 3; k++)
            if (rc != TG_OK || last_status[k] != TG_SOLVED) {
                print_error(std::string("[MPC12STATE::mpcLoop] ") + names[k] + " -  MPC solution is not found");
                return false;
            }
        _x_opt.swap(X);
        _u_opt.swap(U);
        return true;
    }
    // the per-stage update entry points only refresh problem data in the reference (:1357-1372);
    // the engine rebuilds it inside every solve
    bool updateXYMPC() { return _is_MPC_initialized; }
    bool updateZMPC() { return _is_MPC_initialized; }
    bool updateYawMPC() { return _is_MPC_initialized; }

    VectorXd getOptimalStateTraj() { return make_vec(_x_opt.data(), (int)_x_opt.size()); }
    VectorXd getOptimalControlTraj() { return make_vec(_u_opt.data(), (int)_u_opt.size()); }
    int getNumOfStates() { return NUM_OF_STATES; }
    int getNumOfInputs() { return NUM_OF_INPUTS; }
    double getXYMaxVel() { return _p.xy_max_vel; }
    double getZMaxVel() { return _p.z_max_vel; }
    double getYawMaxVel() { return _p.yaw_max_vel; }
    double getXYMaxAccel() { return _p.xy_max_accel; }
    double getZMaxAccel() { return _p.z_max_accel; }
    double getYawMaxAccel() { return _p.yaw_max_accel; }
    double getXYMaxJerk() { return _p.xy_max_jerk; }
    double getZMaxJerk() { return _p.z_max_jerk; }
    double getYawMaxJerk() { return _p.yaw_max_jerk; }

    MatrixXd getXYTransitionMatrix() { return stage_mat(TG_FAM_XY, TG_MAT_AD, 6, 6); }
    MatrixXd getZTransitionMatrix() { return stage_mat(TG_FAM_Z, TG_MAT_AD, 3, 3); }
    MatrixXd getYawTransitionMatrix() { return stage_mat(TG_FAM_YAW, TG_MAT_AD, 3, 3); }
    MatrixXd getXYInputMatrix() { return stage_mat(TG_FAM_XY, TG_MAT_BD, 6, 2); }
    MatrixXd getZInputMatrix() { return stage_mat(TG_FAM_Z, TG_MAT_BD, 3, 1); }
    MatrixXd getYawInputMatrix() { return stage_mat(TG_FAM_YAW, TG_MAT_BD, 3, 1); }
    MatrixXd getXYQ() { return stage_mat(TG_FAM_XY, TG_MAT_Q, 6, 6); }
    MatrixXd getZQ() { return stage_mat(TG_FAM_Z, TG_MAT_Q, 3, 3); }
    MatrixXd getYawQ() { return stage_mat(TG_FAM_YAW, TG_MAT_Q, 3, 3); }
    MatrixXd getXYR() { return stage_mat(TG_FAM_XY, TG_MAT_R, 2, 2); }
    MatrixXd getZR() { return stage_mat(TG_FAM_Z, TG_MAT_R, 1, 1); }
    MatrixXd getYawR() { return stage_mat(TG_FAM_YAW, TG_MAT_R, 1, 1); }
    MatrixXd getXYHessianMatrix() { return stage_sq(TG_FAM_XY, TG_MAT_P); }
    MatrixXd getZHessianMatrix() { return stage_sq(TG_FAM_Z, TG_MAT_P); }
    MatrixXd getYawHessianMatrix() { return stage_sq(TG_FAM_YAW, TG_MAT_P); }
    MatrixXd getXYContraintsMatrix() { return stage_A(TG_FAM_XY); }
    MatrixXd getZContraintsMatrix() { return stage_A(TG_FAM_Z); }
    MatrixXd getYawContraintsMatrix() { return stage_A(TG_FAM_YAW); }
    // 12 x 12 / 12 x 4 block-diagonal assemblies used by the harness roll-out (:2001, :2014)
    MatrixXd getTransitionMatrix() { return blkdiag(getXYTransitionMatrix(), getZTransitionMatrix(), getYawTransitionMatrix()); }
    MatrixXd getInputMatrix() { return blkdiag(getXYInputMatrix(), getZInputMatrix(), getYawInputMatrix()); }
    MatrixXd getQ() { return blkdiag(getXYQ(), getZQ(), getYawQ()); }
    // the reference sizes this one with Q's dimensions (:2051); the 4 x 4 block diagonal is returned
    MatrixXd getR() { return blkdiag(getXYR(), getZR(), getYawR()); }
    // q, l, u of a stage for the current state / reference and the last solution
    VectorXd getXYGradient() { return stage_vec(TG_FAM_XY, 0); }
    VectorXd getZGradient() { return stage_vec(TG_FAM_Z, 0); }
    VectorXd getYawGradient() { return stage_vec(TG_FAM_YAW, 0); }
    VectorXd getXYLowerBounds() { return stage_vec(TG_FAM_XY, 1); }
    VectorXd getZLowerBounds() { return stage_vec(TG_FAM_Z, 1); }
    VectorXd getYawLowerBounds() { return stage_vec(TG_FAM_YAW, 1); }
    VectorXd getXYUpperBounds() { return stage_vec(TG_FAM_XY, 2); }
    VectorXd getZUpperBounds() { return stage_vec(TG_FAM_Z, 2); }
    VectorXd getYawUpperBounds() { return stage_vec(TG_FAM_YAW, 2); }

    // mpc_12state.cpp:2107-2187: the file scripts/plot_trajectories.py reads.  A header, one row
    // for the current state (29 fields, no limits), then N rows; as in the reference, row i+1
    // carries time (i+1)*dt with state block i of the optimal trajectory and reference block
    // i+1 (:2132-2162), the yaw reference being the one computeYawRefTrajectory produced.
    void saveMPCSolutionsToFile()
    {
        std::ofstream f(_outputCSVFile);
        if (!f.is_open()) { print_error("[MPCROS::saveMPCSolutionsToFile] Coudl not open file " + _outputCSVFile); return; }
        f << "time,x,v_x,a_x,y,v_y,a_y,z,v_z,a_z,yaw,v_yaw,a_yaw,j_x,j_y,j_z,j_yaw,des_x,des_vx,des_ax,"
             "des_y,des_vy,des_ay,des_z,des_vz,des_az,des_yaw,des_v_yaw,des_a_yaw,xy_VelMin,xy_VelMax,"
             "xy_AccelMin,xy_AccelMax,xy_JerkMin,xy_JerkMax,z_VelMin,z_VelMax,z_AccelMin,z_AccelMax,"
             "z_JerkMin,z_JerkMax,yaw_VelMin,yaw_VelMax,yaw_AccelMin,yaw_AccelMax,yaw_JerkMin,yaw_JerkMax\n";
        const int N = _mpcWindow;
        std::vector<double> yaw_ref((size_t)N + 1, 0.0);
        const double *d_yaw = nullptr;
        int ld = 0;
        if (_h && last_status[2] != TG_UNSOLVED &&
            tg_mpc12_intermediates(_h, nullptr, nullptr, &d_yaw, &ld) == TG_OK && d_yaw)
            for (int i = 0; i <= N; i++) tg_copy_to_host(&yaw_ref[(size_t)i], d_yaw + (size_t)i * ld, sizeof(double));
        const double lim[18] = {-_p.xy_max_vel, _p.xy_max_vel, -_p.xy_max_accel, _p.xy_max_accel, -_p.xy_max_jerk, _p.xy_max_jerk,
                                -_p.z_max_vel, _p.z_max_vel, -_p.z_max_accel, _p.z_max_accel, -_p.z_max_jerk, _p.z_max_jerk,
                                -_p.yaw_max_vel, _p.yaw_max_vel, -_p.yaw_max_accel, _p.yaw_max_accel, -_p.yaw_max_jerk, _p.yaw_max_jerk};
        f << "0.0";
        for (int k = 0; k < 12; k++) f << "," << _current_state[(size_t)k];
        for (int k = 0; k < NUM_OF_INPUTS; k++) f << ",0.0";
        for (int k = 0; k < 12; k++) f << "," << _referenceTraj[(size_t)k];
        f << "\n";
        for (int i = 0; i < N; i++) {
            f << (i + 1) * _dt;
            for (int k = 0; k < 12; k++) f << "," << _x_opt[(size_t)(12 * i + k)];
            for (int k = 0; k < 4; k++) f << "," << _u_opt[(size_t)(4 * i + k)];
            for (int k = 0; k < 9; k++) f << "," << _referenceTraj[(size_t)(12 * (i + 1) + k)];
            f << "," << yaw_ref[(size_t)i + 1] << ",0,0";
            for (int k = 0; k < 18; k++) f << "," << lim[k];
            f << "\n";
        }
        print_info("[MPC12STATE::saveMPCSolutionsToFile] Saved MPC solutions to file: " + _outputCSVFile);
    }

    void setSolverOptions(const tg_solver_opts &o) { _opts = o; _opts_set = true; }
    int last_status[3] = {TG_UNSOLVED, TG_UNSOLVED, TG_UNSOLVED}, last_iters[3] = {0, 0, 0};   // {z, xy, yaw}

private:
    static bool weight(double w, double &dst, const char *name)
    {
        if (w < 0.0) { print_error(std::string(name) + " < 0"); return false; }
        dst = w;
        return true;
    }
    bool limit(double v, double &dst, int bit, const char *name, double fallback)
    {
        _limits_set |= 1u << bit;
        if (v < 0) { print_error(std::string(name) + " < 0. Setting default"); dst = fallback; return false; }
        dst = v;
        return true;
    }
    MatrixXd stage_mat(int stage, int which, int r, int c)
    {
        std::vector<double> b((size_t)r * c, 0.0);
        if (_h) tg_mpc12_get(_h, stage, which, b.data());
        return make_mat(b.data(), r, c);
    }
    MatrixXd stage_sq(int stage, int which)
    {
        int n = 0, m = 0;
        if (_h) tg_mpc12_dims(_h, stage, &n, &m);
        return stage_mat(stage, which, n, n);
    }
    MatrixXd stage_A(int stage)
    {
        int n = 0, m = 0;
        if (_h) tg_mpc12_dims(_h, stage, &n, &m);
        return stage_mat(stage, TG_MAT_A, m, n);
    }
    VectorXd stage_vec(int stage, int which)
    {
        int n = 0, m = 0;
        if (!_h) return VectorXd();
        tg_mpc12_dims(_h, stage, &n, &m);
        std::vector<double> q((size_t)n), l((size_t)m), u((size_t)m);
        tg_mpc12_assemble_host(_h, stage, 1, _current_state.data(), _referenceTraj.data(), _x_opt.data(), q.data(),
                               l.data(), u.data());
        const std::vector<double> &o = which == 0 ? q : (which == 1 ? l : u);
        return make_vec(o.data(), (int)o.size());
    }
    static MatrixXd blkdiag(const MatrixXd &a, const MatrixXd &b, const MatrixXd &c)
    {
        const MatrixXd *ms[3] = {&a, &b, &c};
        int r = 0, cc = 0;
        for (auto *m : ms) { r += (int)m->rows(); cc += (int)m->cols(); }
        std::vector<double> buf((size_t)r * cc, 0.0);
        int i0 = 0, j0 = 0;
        for (auto *m : ms) {
            for (int i = 0; i < (int)m->rows(); i++)
                for (int j = 0; j < (int)m->cols(); j++) buf[(size_t)(i0 + i) * cc + j0 + j] = (*m)(i, j);
            i0 += (int)m->rows(); j0 += (int)m->cols();
        }
        return make_mat(buf.data(), r, cc);
    }
    bool _debug = false, _state_received = false, _received_refTraj = false, _enable_control_smoothing = false,
         _is_MPC_initialized = false;
    double _dt = 0.05, _alt_above_target = 1.0, _minAltitude = 1.0;
    int _mpcWindow = 20, _device = 0;
    unsigned _limits_set = 0;
    tg_mpc12_params _p = [] {
        tg_mpc12_params p{};
        p.xy_state_weight = p.z_state_weight = p.yaw_state_weight = 1.0;
        p.xy_input_weight = p.z_input_weight = p.yaw_input_weight = 0.1;
        p.xy_smooth_input_weight = p.z_smooth_input_weight = p.yaw_smooth_input_weight = 10.0;
        return p;
    }();
    std::vector<double> _current_state = std::vector<double>(12, 0.0), _referenceTraj, _x_opt, _u_opt;
    std::string _outputCSVFile;
    tg_mpc12 *_h = nullptr;
    tg_solver_opts _opts{};
    bool _opts_set = false;
};

} // namespace tg
