"""Minimal ROS2 stand-ins so that the reference's Python service node can be
imported BY PATH inside the development container (no ROS2 there).

TEST INFRASTRUCTURE ONLY.  Nothing in the product path imports this file.  It is
used by ``oracle/gen_golden.py`` (run in the development container, where
``/root/reference`` exists) to produce the golden vectors under
``tests/golden/``; the GPU box never sees the reference.

What has to be faked is listed in SURVEY.md §8c "Stub recipe":
  rclpy / rclpy.node / rclpy.parameter / rclpy.time, neo_srvs2.srv,
  geometry_msgs.msg, nav_msgs.msg, neo_nav2_py_costmap2D.{line_iterator,costmap},
  tf2_ros{,.buffer,.transform_listener}, rcl_interfaces.msg
(reference imports: neo_mpc_planner2/mpc_optimization_server.py:25-42).

The ``Costmap2d`` stand-in implements THIS BUILD's documented costmap contract
(DESIGN.md "Costmap contract"); the real ``neo_nav2_py_costmap2D`` is an
un-vendored, un-pinned dependency of the reference (README.md:22) and is absent,
so parity at that boundary is "unpinned" by construction.
"""
import math
import sys
import types

import numpy as np


# --------------------------------------------------------------------------- messages
class _Msg:
    """Value-compared attribute bag (rclpy messages compare by value and type)."""
    __slots__ = ()

    def __eq__(self, other):
        if type(other) is not type(self):
            return False
        return all(getattr(self, s) == getattr(other, s) for s in self.__slots__)

    def __ne__(self, other):
        return not self.__eq__(other)

    __hash__ = None


class Point(_Msg):
    __slots__ = ("x", "y", "z")

    def __init__(self, x=0.0, y=0.0, z=0.0):
        self.x, self.y, self.z = x, y, z


Point32 = Point
Vector3 = Point


class Quaternion(_Msg):
    __slots__ = ("x", "y", "z", "w")

    def __init__(self, x=0.0, y=0.0, z=0.0, w=1.0):
        self.x, self.y, self.z, self.w = x, y, z, w


class Header(_Msg):
    __slots__ = ("stamp", "frame_id")

    def __init__(self):
        self.stamp, self.frame_id = 0, ""


class Pose(_Msg):
    __slots__ = ("position", "orientation")

    def __init__(self):
        self.position, self.orientation = Point(), Quaternion()


class PoseStamped(_Msg):
    __slots__ = ("header", "pose")

    def __init__(self):
        self.header, self.pose = Header(), Pose()


class Twist(_Msg):
    __slots__ = ("linear", "angular")

    def __init__(self):
        self.linear, self.angular = Vector3(), Vector3()


class TwistStamped(_Msg):
    __slots__ = ("header", "twist")

    def __init__(self):
        self.header, self.twist = Header(), Twist()


class Polygon(_Msg):
    # ``points`` is a plain attribute: assigning another polygon's list aliases it,
    # exactly like the rclpy setter (needed to reproduce SURVEY §8a-4).
    __slots__ = ("points",)

    def __init__(self):
        self.points = []


class PolygonStamped(_Msg):
    __slots__ = ("header", "polygon")

    def __init__(self):
        self.header, self.polygon = Header(), Polygon()


class Path(_Msg):
    __slots__ = ("header", "poses")

    def __init__(self):
        self.header, self.poses = Header(), []


class OccupancyGrid(_Msg):
    __slots__ = ("header", "data")

    def __init__(self):
        self.header, self.data = Header(), []


class _OptimizerRequest(_Msg):
    __slots__ = ("current_pose", "carrot_pose", "goal_pose", "current_vel",
                 "switch_opt", "control_interval")

    def __init__(self):
        self.current_pose = PoseStamped()
        self.carrot_pose = PoseStamped()
        self.goal_pose = Pose()
        self.current_vel = Twist()
        self.switch_opt = False
        self.control_interval = 0.0


class _OptimizerResponse(_Msg):
    __slots__ = ("output_vel",)

    def __init__(self):
        self.output_vel = TwistStamped()


class Optimizer:
    Request = _OptimizerRequest
    Response = _OptimizerResponse


# --------------------------------------------------------------------------- costmap
def nav2_occupancy_table():
    """nav2 Costmap2DPublisher translation raw u8 -> occupancy [-1, 100].
    (0 -> 0, 253 -> 99, 254 -> 100, 255 -> -1, 1..252 -> 1 + 97*(v-1)/251 int div)"""
    t = np.zeros(256, dtype=np.int64)
    for v in range(1, 253):
        t[v] = 1 + (97 * (v - 1)) // 251
    t[253], t[254], t[255] = 99, 100, -1
    return t


def bresenham_cells(x0, y0, x1, y1):
    """Integer Bresenham line, end points inclusive (build's own rasteriser contract)."""
    cells = []
    dx, dy = abs(x1 - x0), abs(y1 - y0)
    sx = 1 if x1 >= x0 else -1
    sy = 1 if y1 >= y0 else -1
    err = dx - dy
    x, y = x0, y0
    while True:
        cells.append((x, y))
        if x == x1 and y == y1:
            break
        e2 = 2 * err
        if e2 > -dy:
            err -= dy
            x += sx
        if e2 < dx:
            err += dx
            y += sy
    return cells


class Costmap2d:
    """Stand-in for neo_nav2_py_costmap2D.costmap.Costmap2d implementing the BUILD's
    costmap contract.  A class-level ``pending`` map is picked up by the instance the
    reference constructs at mpc_optimization_server.py:118."""
    pending = None  # (cells u8 [size_y, size_x], resolution, origin_x, origin_y)

    def __init__(self, node=None):
        self.table = nav2_occupancy_table()
        if Costmap2d.pending is not None:
            self.set_map(*Costmap2d.pending)
        else:
            self.set_map(np.zeros((1, 1), np.uint8), 1.0, 0.0, 0.0)

    def set_map(self, cells, resolution, origin_x, origin_y):
        self.cells = np.ascontiguousarray(cells, dtype=np.uint8)
        self.size_y, self.size_x = self.cells.shape
        self.resolution = float(resolution)
        self.origin_x = float(origin_x)
        self.origin_y = float(origin_y)

    def getWorldToMap(self, wx, wy):
        mx = int(math.floor((float(wx) - self.origin_x) / self.resolution))
        my = int(math.floor((float(wy) - self.origin_y) / self.resolution))
        return mx, my

    def getCost(self, mx, my):
        if mx < 0 or my < 0 or mx >= self.size_x or my >= self.size_y:
            return 1.0
        return float(self.table[self.cells[my, mx]]) / 100.0

    def getFootprintCost(self, polygon):
        pts = polygon.points
        n = len(pts)
        if n == 0:
            return 0.0
        cells = [self.getWorldToMap(p.x, p.y) for p in pts]
        worst = -1.0
        for i in range(n):
            (x0, y0), (x1, y1) = cells[i], cells[(i + 1) % n]
            for (mx, my) in bresenham_cells(x0, y0, x1, y1):
                c = self.getCost(mx, my)
                if c > worst:
                    worst = c
        return worst


class LineIterator:  # imported but never used by the reference (py:35)
    pass


# --------------------------------------------------------------------------- rclpy
class _Param:
    def __init__(self, value):
        self.value = value


class _Logger:
    def info(self, *a, **k):
        pass

    warn = error = debug = info


class _Stamp:
    def to_msg(self):
        return 0


class _Clock:
    def now(self):
        return _Stamp()


class _Pub:
    def __init__(self):
        self.last = None

    def publish(self, msg):
        self.last = msg


class Node:
    #: parameter overrides applied at declare time (the launch-file YAML stand-in)
    overrides = {}

    def __init__(self, name):
        self._name = name
        self._params = {}

    def declare_parameter(self, name, value=None):
        self._params[name] = _Param(Node.overrides.get(name, value))
        return self._params[name]

    def get_parameter(self, name):
        return self._params[name]

    def create_service(self, *a, **k):
        return object()

    def create_publisher(self, *a, **k):
        return _Pub()

    def create_subscription(self, *a, **k):
        return object()

    def add_on_set_parameters_callback(self, cb):
        self._param_cb = cb

    def get_logger(self):
        return _Logger()

    def get_clock(self):
        return _Clock()


class TransformException(Exception):
    pass


class _Vec3:
    def __init__(self, x=0.0, y=0.0, z=0.0):
        self.x, self.y, self.z = x, y, z


class _Transform:
    def __init__(self, x, y, q):
        self.translation = _Vec3(x, y, 0.0)
        self.rotation = Quaternion()
        self.rotation.x, self.rotation.y, self.rotation.z, self.rotation.w = q


class TransformStamped:
    def __init__(self, x, y, q):
        self.header, self.transform = Header(), _Transform(x, y, q)


class Buffer:
    #: (x, y, (qx, qy, qz, qw)) = the map -> base_link transform `lookup_transform` answers with
    #: (py:275-278).  None: no tf -- publishLocalPlan returns early (py:279-282).
    pending = None

    def lookup_transform(self, *a, **k):
        if Buffer.pending is None:
            raise TransformException("no tf in the oracle harness")
        x, y, q = Buffer.pending
        return TransformStamped(float(x), float(y), tuple(float(v) for v in q))


class TransformListener:
    def __init__(self, *a, **k):
        pass


class SetParametersResult:
    def __init__(self, successful=True):
        self.successful = successful


class Parameter:
    class Type:
        DOUBLE = 3

    def __init__(self, name, type_=3, value=None):
        self.name, self.type_, self.value = name, type_, value


def install():
    """Register the stub modules in ``sys.modules``."""
    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    rclpy = mod("rclpy", init=lambda args=None: None, spin=lambda node: None,
                shutdown=lambda: None)
    rclpy.node = mod("rclpy.node", Node=Node)
    rclpy.parameter = mod("rclpy.parameter", Parameter=Parameter)
    rclpy.time = mod("rclpy.time", Time=lambda *a, **k: 0)
    mod("neo_srvs2")
    mod("neo_srvs2.srv", Optimizer=Optimizer)
    mod("geometry_msgs")
    mod("geometry_msgs.msg", TwistStamped=TwistStamped, PoseStamped=PoseStamped, Pose=Pose,
        Polygon=Polygon, PolygonStamped=PolygonStamped, Twist=Twist, Point=Point,
        Point32=Point32, Quaternion=Quaternion)
    mod("nav_msgs")
    mod("nav_msgs.msg", OccupancyGrid=OccupancyGrid, Path=Path)
    mod("neo_nav2_py_costmap2D")
    mod("neo_nav2_py_costmap2D.line_iterator", LineIterator=LineIterator)
    mod("neo_nav2_py_costmap2D.costmap", Costmap2d=Costmap2d)
    tf2 = mod("tf2_ros", TransformException=TransformException)
    tf2.buffer = mod("tf2_ros.buffer", Buffer=Buffer)
    tf2.transform_listener = mod("tf2_ros.transform_listener",
                                 TransformListener=TransformListener)
    mod("rcl_interfaces")
    mod("rcl_interfaces.msg", SetParametersResult=SetParametersResult)


def load_reference(path="/root/reference/neo_mpc_planner2/mpc_optimization_server.py"):
    """Import the reference node module by path (development container only)."""
    import importlib.util
    install()
    spec = importlib.util.spec_from_file_location("_neo_ref_mpc_server", path)
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m
