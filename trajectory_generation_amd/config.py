"""Loader for the reference's parameter files (config/mpc.yaml, config/mpc_12state.yaml): the
same key names the ROS nodes declare (mpc_ros.cpp:39-77, mpc_12state_ros.cpp:39-114), applied to
the MPC / MPC12STATE look-alikes in the same order the nodes' constructors apply them."""
import yaml


def _ros_params(path, node=None):
    with open(path) as f:
        doc = yaml.load(f, Loader=yaml.SafeLoader)
    if not isinstance(doc, dict) or not doc:
        raise ValueError("%s: not a ROS 2 parameter file" % path)
    if node is None:
        node = next(iter(doc))
    try:
        return dict(doc[node]["ros__parameters"])
    except (KeyError, TypeError):
        raise ValueError("%s: no %s/ros__parameters section" % (path, node))


def configure_mpc6(mpc, path, node=None):
    """MPCROS::MPCROS (mpc_ros.cpp:37-84): debug, dt_pred, mpc_window, state / input / smoothing
    weights, minimum_altitude, max_velocity, max_acceleration.  Returns the parameter dict; False
    returns of the setters are collected under '_rejected'."""
    p = _ros_params(path, node)
    rejected = []

    def call(key, fn, conv=lambda v: v):
        if key in p and fn(conv(p[key])) is False:
            rejected.append(key)

    if "debug" in p:
        mpc.setDebug(bool(p["debug"]))
    call("dt_pred", mpc.setDt, float)
    call("mpc_window", mpc.setMPCWindow, int)
    call("state_weight", mpc.setStateWeight, float)
    call("input_weight", mpc.setInputWeight, float)
    call("smooth_input_weight", mpc.setSmoothInputWeight, float)
    if "enable_control_smoothing" in p:
        mpc.enableControlSmoothing(bool(p["enable_control_smoothing"]))
    call("minimum_altitude", mpc.setMinimumAltitude, float)
    call("max_velocity", mpc.setMaxVel, lambda v: [float(x) for x in v])
    call("max_acceleration", mpc.setMaxAccel, lambda v: [float(x) for x in v])
    p["_rejected"] = rejected
    return p


def configure_mpc12(mpc, path, node=None):
    """MPCROS::MPCROS of the 12-state node (mpc_12state_ros.cpp:37-121)."""
    p = _ros_params(path, node)
    rejected = []

    def call(key, fn, conv=float):
        if key in p and fn(conv(p[key])) is False:
            rejected.append(key)

    if "debug" in p:
        mpc.setDebug(bool(p["debug"]))
    call("dt_pred", mpc.setDt)
    call("mpc_window", mpc.setMPCWindow, int)
    for ax, cls in (("xy", "XY"), ("z", "Z"), ("yaw", "Yaw")):
        call("%s_state_weight" % ax, getattr(mpc, "set%sStateWeight" % cls))
        call("%s_input_weight" % ax, getattr(mpc, "set%sInputWeight" % cls))
        call("%s_smooth_input_weight" % ax, getattr(mpc, "set%sSmoothInputWeight" % cls))
        call("%s_max_velocity" % ax, getattr(mpc, "set%sMaxVel" % cls))
        call("%s_max_acceleration" % ax, getattr(mpc, "set%sMaxAccel" % cls))
        call("%s_max_jerk" % ax, getattr(mpc, "set%sMaxJerk" % cls))
    if "enable_control_smoothing" in p:
        mpc.enableControlSmoothing(bool(p["enable_control_smoothing"]))
    call("minimum_altitude", mpc.setMinimumAltitude)
    p["_rejected"] = rejected
    return p
