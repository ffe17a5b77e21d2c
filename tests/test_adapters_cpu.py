"""CPU tests of the pieces either side of the hot path: the parameter-file loader with the
reference's key names and the message-shaped view of a solution."""
import numpy as np


def _write(tmp_path, name, text):
    p = tmp_path / name
    p.write_text(text)
    return str(p)


def test_yaml_loader_6dof_keys(tmp_path):
    from trajectory_generation_amd.config import configure_mpc6
    from trajectory_generation_amd.mpc_6dof import MPC
    path = _write(tmp_path, "mpc.yaml", """
mpc_node:
  ros__parameters:
    debug: False
    dt_pred: 0.1
    mpc_window: 5
    minimum_altitude: -1.0
    enable_control_smoothing: False
    smooth_input_weight: 100.0
    max_velocity: [10.0, 10.0, 10.0]
    max_acceleration: [5.0, 5.0, 5.0]
    max_jerk: [2.0, 2.0, 2.0]
    state_weight: 7000.0
    input_weight: 500.0
""")
    m = MPC()
    p = configure_mpc6(m, path)
    assert p["_rejected"] == []
    assert (m._dt, m._mpcWindow, m._state_weight, m._input_weight) == (0.1, 5, 7000.0, 500.0)
    assert m._smooth_input_weight == 100.0 and m._enable_control_smoothing is False and m._minAltitude == -1.0
    assert list(m._maxVel) == [10.0] * 3 and list(m._maxAccel) == [5.0] * 3 and m._debug is False


def test_yaml_loader_12state_keys_and_rejections(tmp_path, capsys):
    from trajectory_generation_amd.config import configure_mpc12
    from trajectory_generation_amd.mpc_12state import MPC12STATE
    path = _write(tmp_path, "mpc_12state.yaml", """
mpc_12state_node:
  ros__parameters:
    dt_pred: 0.1
    mpc_window: 10
    xy_max_velocity: 12.0
    z_max_velocity: 7.0
    yaw_max_velocity: 2.0
    xy_max_acceleration: 6.0
    z_max_acceleration: 10.0
    yaw_max_acceleration: 10.0
    xy_max_jerk: 30.0
    z_max_jerk: -40.0
    yaw_max_jerk: 10.0
    xy_state_weight: 5000.0
    z_state_weight: 7000.0
    yaw_state_weight: 1000.0
    xy_input_weight: 200.0
    z_input_weight: 200.0
    yaw_input_weight: 100.0
    enable_control_smoothing: False
""")
    m = MPC12STATE()
    p = configure_mpc12(m, path)
    assert p["_rejected"] == ["z_max_jerk"] and m._z_MaxJerk == 5.0       # fall-back constant installed
    assert (m._xy_MaxVel, m._z_MaxVel, m._yaw_MaxVel) == (12.0, 7.0, 2.0)
    assert (m._xy_state_weight, m._z_state_weight, m._yaw_state_weight) == (5000.0, 7000.0, 1000.0)
    assert (m._xy_input_weight, m._yaw_input_weight, m._mpcWindow) == (200.0, 100.0, 10)
    assert "[ERROR]" in capsys.readouterr().out


def test_message_mapping_6dof():
    from trajectory_generation_amd.messages import from_mpc6
    N, dt = 4, 0.1
    X = np.arange(6 * (N + 1), dtype=float); U = 100 + np.arange(3 * N, dtype=float)
    m = from_mpc6(X, U, N, dt, start_t=2.0, frame_id="odom", max_vel=(1, 3, 2), max_accel=(5, 4, 6))
    assert len(m.states) == N and len(m.pose_path) == N + 1
    s = m.states[1]                                       # horizon state 2, control of step 1
    assert s.time_from_start == 2 * dt and (s.position.x, s.position.y, s.position.z) == (12.0, 13.0, 14.0)
    assert (s.velocity.x, s.velocity.z) == (15.0, 17.0) and (s.acceleration.x, s.acceleration.z) == (103.0, 105.0)
    assert m.pose_path[0].position.x == 0.0 and m.pose_path[3].stamp == 2.0 + 3 * dt and m.pose_path[0].orientation.w == 1.0
    assert m.max_velocity == 3.0 and m.max_acceleration == 6.0
    assert m.command.translation.x == 6.0 and m.command.acceleration.y == 101.0   # states[0]


def test_message_mapping_12state():
    from trajectory_generation_amd.messages import from_mpc12
    N, dt = 3, 0.05
    X = np.arange(12 * (N + 1), dtype=float) * 0.01; U = 1 + np.arange(4 * N, dtype=float)
    m = from_mpc12(X, U, N, dt)
    s = m.states[0]
    assert (s.position.x, s.velocity.x, s.acceleration.x) == (0.12, 0.13, 0.14)
    assert (s.position.y, s.position.z, s.yaw, s.yaw_speed) == (0.15, 0.18, 0.21, 0.22)
    assert (s.jerk.x, s.jerk.y, s.jerk.z, s.yaw_jerk) == (1.0, 2.0, 3.0, 4.0)
    last = m.states[-1]
    assert m.command.translation.x == last.position.x                      # last horizon state, as the reference
    assert abs(m.command.rotation.z - np.sin(0.5 * last.yaw)) < 1e-15 and abs(m.command.rotation.w - np.cos(0.5 * last.yaw)) < 1e-15
    assert abs(m.pose_path[2].orientation.z - np.sin(0.5 * X[24 + 9])) < 1e-15
