"""
Host-side affine math with Raysect's names and, where results feed the device path (to_local /
to_root matrices, bounding boxes), Raysect's exact f64 operation order — a 1-ulp difference in a
matrix can flip a grazing hit.

Mirrors: raysect/core/math/point.pyx, vector.pyx, normal.pyx, affinematrix.pyx, transform.pyx.
Python floats are IEEE binary64 and CPython never contracts a*b+c into an FMA, so expressions
written in the same order give the same bits as the reference's C.
"""
import math

DEG2RAD = 0.017453292519943295  # transform.pyx:39


class _Vec3:
    __slots__ = ("x", "y", "z")

    def __init__(self, x=0.0, y=0.0, z=0.0):
        self.x = float(x)
        self.y = float(y)
        self.z = float(z)

    def __iter__(self):
        yield self.x
        yield self.y
        yield self.z

    def __getitem__(self, i):
        return (self.x, self.y, self.z)[i]

    def __repr__(self):
        return "%s(%r, %r, %r)" % (type(self).__name__, self.x, self.y, self.z)

    def __eq__(self, other):
        return type(self) is type(other) and self.x == other.x and self.y == other.y and self.z == other.z

    def __hash__(self):
        return hash((type(self).__name__, self.x, self.y, self.z))

    def dot(self, v):
        return self.x * v.x + self.y * v.y + self.z * v.z

    @property
    def length(self):
        return math.sqrt(self.x * self.x + self.y * self.y + self.z * self.z)


def _orthogonal(vec):
    """An arbitrary unit vector orthogonal to `vec` (vector.pyx:440-472 / normal.pyx:346-370): Gram-Schmidt of the x axis (the y axis
    when `vec` lies within 60 degrees of x) against the normalised vector."""
    t = vec.x * vec.x + vec.y * vec.y + vec.z * vec.z
    t = 1.0 / math.sqrt(t)
    nx, ny, nz = vec.x * t, vec.y * t, vec.z * t
    vx, vy, vz = 1.0, 0.0, 0.0
    if abs(nx * vx + ny * vy + nz * vz) > 0.5:
        vx, vy = 0.0, 1.0
    m = nx * vx + ny * vy + nz * vz
    ux, uy, uz = vx - m * nx, vy - m * ny, vz - m * nz
    t = ux * ux + uy * uy + uz * uz
    t = 1.0 / math.sqrt(t)
    return ux * t, uy * t, uz * t


class Vector3D(_Vec3):
    """raysect/core/math/vector.pyx"""
    __slots__ = ()

    def __init__(self, x=0.0, y=0.0, z=1.0):
        super().__init__(x, y, z)

    def __neg__(self):
        return Vector3D(-self.x, -self.y, -self.z)

    def __add__(self, v):
        return Vector3D(self.x + v.x, self.y + v.y, self.z + v.z)

    def __sub__(self, v):
        return Vector3D(self.x - v.x, self.y - v.y, self.z - v.z)

    def __mul__(self, s):
        return Vector3D(s * self.x, s * self.y, s * self.z)

    __rmul__ = __mul__

    def cross(self, v):                                     # vector.pyx:306-310
        return Vector3D(self.y * v.z - v.y * self.z, self.z * v.x - v.z * self.x, self.x * v.y - v.x * self.y)

    def normalise(self):                                    # vector.pyx:313-337
        t = self.x * self.x + self.y * self.y + self.z * self.z
        if t == 0.0:
            raise ZeroDivisionError("A zero length vector can not be normalised as the direction of a zero length vector is undefined.")
        t = 1.0 / math.sqrt(t)
        return Vector3D(self.x * t, self.y * t, self.z * t)

    def transform(self, m):                                 # vector.pyx:339-369
        a = m.m
        return Vector3D(a[0] * self.x + a[1] * self.y + a[2] * self.z,
                        a[4] * self.x + a[5] * self.y + a[6] * self.z,
                        a[8] * self.x + a[9] * self.y + a[10] * self.z)

    def copy(self):
        return Vector3D(self.x, self.y, self.z)

    def neg(self):
        return Vector3D(-self.x, -self.y, -self.z)

    def orthogonal(self):                                   # vector.pyx:440-472
        return Vector3D(*_orthogonal(self))


class Normal3D(_Vec3):
    """raysect/core/math/normal.pyx"""
    __slots__ = ()

    def __init__(self, x=0.0, y=0.0, z=1.0):
        super().__init__(x, y, z)

    def __neg__(self):
        return Normal3D(-self.x, -self.y, -self.z)

    def normalise(self):
        t = self.x * self.x + self.y * self.y + self.z * self.z
        if t == 0.0:
            raise ZeroDivisionError("A zero length vector can not be normalised as the direction of a zero length vector is undefined.")
        t = 1.0 / math.sqrt(t)
        return Normal3D(self.x * t, self.y * t, self.z * t)

    def transform_with_inverse(self, m):                    # normal.pyx:250-271
        a = m.m
        return Normal3D(a[0] * self.x + a[4] * self.y + a[8] * self.z,
                        a[1] * self.x + a[5] * self.y + a[9] * self.z,
                        a[2] * self.x + a[6] * self.y + a[10] * self.z)

    def transform(self, m):                                 # normal.pyx:222-248
        return self.transform_with_inverse(m.inverse())

    def as_vector(self):
        return Vector3D(self.x, self.y, self.z)

    def copy(self):
        return Normal3D(self.x, self.y, self.z)

    def neg(self):
        return Normal3D(-self.x, -self.y, -self.z)

    def cross(self, v):                                     # normal.pyx:200-204
        return Vector3D(self.y * v.z - v.y * self.z, self.z * v.x - v.z * self.x, self.x * v.y - v.x * self.y)

    def orthogonal(self):                                   # normal.pyx:346-370
        return Vector3D(*_orthogonal(self))


class Point3D(_Vec3):
    """raysect/core/math/point.pyx"""
    __slots__ = ()

    def __init__(self, x=0.0, y=0.0, z=0.0):
        super().__init__(x, y, z)

    def __add__(self, v):
        return Point3D(self.x + v.x, self.y + v.y, self.z + v.z)

    def __sub__(self, v):
        return Point3D(self.x - v.x, self.y - v.y, self.z - v.z)

    def vector_to(self, p):
        return Vector3D(p.x - self.x, p.y - self.y, p.z - self.z)

    def distance_to(self, p):
        x, y, z = p.x - self.x, p.y - self.y, p.z - self.z
        return math.sqrt(x * x + y * y + z * z)

    def transform(self, m):                                 # point.pyx:253-284
        a = m.m
        w = a[12] * self.x + a[13] * self.y + a[14] * self.z + a[15]
        if w == 0.0:
            raise ZeroDivisionError("Bad matrix transform, 4th element of homogeneous coordinate is zero.")
        w = 1.0 / w
        return Point3D((a[0] * self.x + a[1] * self.y + a[2] * self.z + a[3]) * w,
                       (a[4] * self.x + a[5] * self.y + a[6] * self.z + a[7]) * w,
                       (a[8] * self.x + a[9] * self.y + a[10] * self.z + a[11]) * w)

    def copy(self):
        return Point3D(self.x, self.y, self.z)


# 2x2 minors used by the reference's Cramer inverse: t[k] = m[r][a]*m[s][b] - m[r][b]*m[s][a]
_MINORS_01 = ((0, 1), (0, 2), (0, 3), (1, 2), (1, 3), (2, 3))


class AffineMatrix3D:
    """raysect/core/math/affinematrix.pyx — row-major 4x4 f64, stored flat in ``m`` (16 floats)."""
    __slots__ = ("m",)

    def __init__(self, m=None):
        if m is None:
            self.m = [1.0, 0.0, 0.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0, 0.0, 1.0]
        elif isinstance(m, AffineMatrix3D):
            self.m = list(m.m)
        else:
            rows = [list(r) for r in m] if hasattr(m[0], "__len__") else [list(m[i * 4:i * 4 + 4]) for i in range(4)]
            if len(rows) != 4 or any(len(r) != 4 for r in rows):
                raise TypeError("AffineMatrix3D requires a 4x4 matrix.")
            self.m = [float(v) for r in rows for v in r]

    @classmethod
    def _new(cls, *v):
        o = cls.__new__(cls)
        o.m = [float(x) for x in v]
        return o

    def __getitem__(self, ij):
        i, j = ij
        if not (0 <= i < 4 and 0 <= j < 4):
            raise IndexError("Indices are out of range.")
        return self.m[i * 4 + j]

    def __setitem__(self, ij, v):
        i, j = ij
        self.m[i * 4 + j] = float(v)

    def __repr__(self):
        r = [self.m[i * 4:i * 4 + 4] for i in range(4)]
        return "AffineMatrix3D(%r)" % (r,)

    def __eq__(self, o):
        return isinstance(o, AffineMatrix3D) and self.m == o.m

    def __mul__(self, o):
        if isinstance(o, AffineMatrix3D):
            return self.mul(o)
        if isinstance(o, (Point3D, Vector3D, Normal3D)):
            return o.transform(self)
        return NotImplemented

    def mul(self, o):                                       # affinematrix.pyx:255-273 (row . column, k ascending)
        a, b = self.m, o.m
        out = [0.0] * 16
        for i in range(4):
            r = i * 4
            for j in range(4):
                out[r + j] = a[r] * b[j] + a[r + 1] * b[4 + j] + a[r + 2] * b[8 + j] + a[r + 3] * b[12 + j]
        return AffineMatrix3D._new(*out)

    def inverse(self):                                      # affinematrix.pyx:172-253 (Cramer's rule, 22 cached minors)
        m = self.m

        def e(r, c):
            return m[r * 4 + c]

        # t[0..5]: rows (0,1); t[6..11]: rows (0,3); t[12..17]: rows (1,3)
        t = [0.0] * 22
        for k, (a, b) in enumerate(_MINORS_01):
            t[k] = e(0, a) * e(1, b) - e(0, b) * e(1, a)
        t[18] = e(2, 0) * t[3] - e(2, 1) * t[1] + e(2, 2) * t[0]
        t[19] = e(2, 0) * t[4] - e(2, 1) * t[2] + e(2, 3) * t[0]
        t[20] = e(2, 0) * t[5] - e(2, 2) * t[2] + e(2, 3) * t[1]
        t[21] = e(2, 1) * t[5] - e(2, 2) * t[4] + e(2, 3) * t[3]
        det = t[20] * e(3, 1) + t[18] * e(3, 3) - t[21] * e(3, 0) - t[19] * e(3, 2)
        if abs(det) < 1e-14:
            raise ValueError("Matrix is singular and not invertible.")
        idet = 1.0 / det
        for k, (a, b) in enumerate(_MINORS_01):
            t[6 + k] = e(0, a) * e(3, b) - e(0, b) * e(3, a)
            t[12 + k] = e(1, a) * e(3, b) - e(1, b) * e(3, a)
        return AffineMatrix3D._new(
            (e(2, 2) * t[16] - e(2, 1) * t[17] - e(2, 3) * t[15]) * idet,
            (e(2, 1) * t[11] - e(2, 2) * t[10] + e(2, 3) * t[9]) * idet,
            (e(3, 1) * t[5] - e(3, 2) * t[4] + e(3, 3) * t[3]) * idet,
            -t[21] * idet,
            (e(2, 0) * t[17] - e(2, 2) * t[14] + e(2, 3) * t[13]) * idet,
            (e(2, 2) * t[8] - e(2, 0) * t[11] - e(2, 3) * t[7]) * idet,
            (e(3, 2) * t[2] - e(3, 0) * t[5] - e(3, 3) * t[1]) * idet,
            t[20] * idet,
            (e(2, 1) * t[14] - e(2, 0) * t[16] - e(2, 3) * t[12]) * idet,
            (e(2, 0) * t[10] - e(2, 1) * t[8] + e(2, 3) * t[6]) * idet,
            (e(3, 0) * t[4] - e(3, 1) * t[2] + e(3, 3) * t[0]) * idet,
            -t[19] * idet,
            (e(2, 0) * t[15] - e(2, 1) * t[13] + e(2, 2) * t[12]) * idet,
            (e(2, 1) * t[7] - e(2, 0) * t[9] - e(2, 2) * t[6]) * idet,
            (e(3, 1) * t[1] - e(3, 0) * t[3] - e(3, 2) * t[0]) * idet,
            t[18] * idet)

    def copy(self):
        return AffineMatrix3D._new(*self.m)


def translate(x, y, z):                                     # transform.pyx:42-72
    return AffineMatrix3D._new(1, 0, 0, x, 0, 1, 0, y, 0, 0, 1, z, 0, 0, 0, 1)


def rotate_x(angle):                                        # transform.pyx:75-106
    r = DEG2RAD * angle
    return AffineMatrix3D._new(1, 0, 0, 0, 0, math.cos(r), -math.sin(r), 0, 0, math.sin(r), math.cos(r), 0, 0, 0, 0, 1)


def rotate_y(angle):                                        # transform.pyx:109-134
    r = DEG2RAD * angle
    return AffineMatrix3D._new(math.cos(r), 0, math.sin(r), 0, 0, 1, 0, 0, -math.sin(r), 0, math.cos(r), 0, 0, 0, 0, 1)


def rotate_z(angle):                                        # transform.pyx:137-162
    r = DEG2RAD * angle
    return AffineMatrix3D._new(math.cos(r), -math.sin(r), 0, 0, math.sin(r), math.cos(r), 0, 0, 0, 0, 1, 0, 0, 0, 0, 1)


def rotate_vector(angle, v):                                # transform.pyx:165-205
    vn = v.normalise()
    r = DEG2RAD * angle
    s, c = math.sin(r), math.cos(r)
    ci = 1.0 - c
    return AffineMatrix3D._new(
        vn.x * vn.x + (1.0 - vn.x * vn.x) * c, vn.x * vn.y * ci - vn.z * s, vn.x * vn.z * ci + vn.y * s, 0,
        vn.x * vn.y * ci + vn.z * s, vn.y * vn.y + (1.0 - vn.y * vn.y) * c, vn.y * vn.z * ci - vn.x * s, 0,
        vn.x * vn.z * ci - vn.y * s, vn.y * vn.z * ci + vn.x * s, vn.z * vn.z + (1.0 - vn.z * vn.z) * c, 0,
        0, 0, 0, 1)


def rotate(yaw, pitch, roll):                               # transform.pyx:208-219
    return rotate_y(-yaw) * rotate_x(-pitch) * rotate_z(roll)
