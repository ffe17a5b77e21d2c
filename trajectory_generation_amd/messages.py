"""ROS-free, message-shaped view of a solution: the field mapping of MPCROS::extractSolution
(mpc_ros.cpp:343-409 for the 6-DOF node, mpc_12state_ros.cpp:410-501 for the 12-state node) as
plain dataclasses, so a ROS node can be re-attached by copying fields one to one.

As in the reference, states[i] of the trajectory message is horizon state i+1 with
time_from_start = (i+1) dt and carries the control of step i (acceleration for the 6-DOF model,
jerk for the 12-state model); the pose path holds all N+1 predicted positions."""
import math
from dataclasses import dataclass, field
from typing import List


@dataclass
class Vec3:
    x: float = 0.0
    y: float = 0.0
    z: float = 0.0


@dataclass
class Quat:
    x: float = 0.0
    y: float = 0.0
    z: float = 0.0
    w: float = 1.0


@dataclass
class StateTrajectoryPoint:
    time_from_start: float = 0.0
    position: Vec3 = field(default_factory=Vec3)
    velocity: Vec3 = field(default_factory=Vec3)
    acceleration: Vec3 = field(default_factory=Vec3)
    jerk: Vec3 = field(default_factory=Vec3)
    yaw: float = 0.0
    yaw_speed: float = 0.0
    yaw_acceleration: float = 0.0
    yaw_jerk: float = 0.0


@dataclass
class PoseStamped:
    stamp: float = 0.0
    frame_id: str = ""
    position: Vec3 = field(default_factory=Vec3)
    orientation: Quat = field(default_factory=Quat)


@dataclass
class MultiDofPoint:
    translation: Vec3 = field(default_factory=Vec3)
    rotation: Quat = field(default_factory=Quat)
    velocity: Vec3 = field(default_factory=Vec3)
    acceleration: Vec3 = field(default_factory=Vec3)


@dataclass
class SolutionMessages:
    stamp: float = 0.0
    frame_id: str = ""
    states: List[StateTrajectoryPoint] = field(default_factory=list)   # StateTrajectory.states
    max_velocity: float = 0.0
    max_acceleration: float = 0.0
    pose_path: List[PoseStamped] = field(default_factory=list)          # nav_msgs/Path
    command: MultiDofPoint = field(default_factory=MultiDofPoint)       # MultiDOFJointTrajectory point


def _yaw_quat(yaw):
    return Quat(0.0, 0.0, math.sin(0.5 * yaw), math.cos(0.5 * yaw))   # tf2 setRPY(0, 0, yaw)


def from_mpc6(X, U, N, dt, start_t=0.0, frame_id="", max_vel=(0, 0, 0), max_accel=(0, 0, 0)):
    """mpc_ros.cpp:343-409.  X [6(N+1)], U [3N] as returned by getOptimalStateTraj / ControlTraj."""
    m = SolutionMessages(stamp=start_t, frame_id=frame_id, max_velocity=float(max(max_vel)),
                         max_acceleration=float(max(max_accel)))
    for i in range(N + 1):
        m.pose_path.append(PoseStamped(start_t + i * dt, frame_id, Vec3(X[6 * i], X[6 * i + 1], X[6 * i + 2]), Quat()))
        if i < N:
            j = 6 * (i + 1)
            m.states.append(StateTrajectoryPoint(
                time_from_start=(i + 1) * dt, position=Vec3(X[j], X[j + 1], X[j + 2]),
                velocity=Vec3(X[j + 3], X[j + 4], X[j + 5]), acceleration=Vec3(U[3 * i], U[3 * i + 1], U[3 * i + 2])))
    s0 = m.states[0]   # first control and the state it leads to: what a low-level controller tracks
    m.command = MultiDofPoint(translation=s0.position, velocity=s0.velocity, acceleration=s0.acceleration)
    return m


def from_mpc12(X, U, N, dt, start_t=0.0, frame_id=""):
    """mpc_12state_ros.cpp:410-501.  X [12(N+1)], U [4N]; the command is taken from the LAST
    horizon state, as the reference does (:476)."""
    m = SolutionMessages(stamp=start_t, frame_id=frame_id)
    for i in range(N + 1):
        b = 12 * i
        m.pose_path.append(PoseStamped(start_t + i * dt, frame_id, Vec3(X[b], X[b + 3], X[b + 6]), _yaw_quat(X[b + 9])))
        if i < N:
            j = 12 * (i + 1)
            m.states.append(StateTrajectoryPoint(
                time_from_start=(i + 1) * dt,
                position=Vec3(X[j], X[j + 3], X[j + 6]), velocity=Vec3(X[j + 1], X[j + 4], X[j + 7]),
                acceleration=Vec3(X[j + 2], X[j + 5], X[j + 8]), jerk=Vec3(U[4 * i], U[4 * i + 1], U[4 * i + 2]),
                yaw=X[j + 9], yaw_speed=X[j + 10], yaw_acceleration=X[j + 11], yaw_jerk=U[4 * i + 3]))
    s = m.states[-1]
    m.command = MultiDofPoint(translation=s.position, rotation=_yaw_quat(s.yaw), velocity=s.velocity,
                              acceleration=s.acceleration)
    return m
