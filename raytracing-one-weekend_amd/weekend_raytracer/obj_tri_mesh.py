"""
OBJ reader (triangles only) with the reference's conventions (reference
obj_tri_mesh.py): faces are lists of three (vertex, uv, normal) 0-based
index tuples, -1 where absent or where the index exceeds the number of
elements read so far; oddities are printed, not raised.

get_smooth_vertex_normal is linear in the mesh size: face normals and the
vertex -> faces adjacency are built once per read, and each vertex normal is
formed from them with the same numpy calls the reference makes, so the
float32 results are identical.
"""
import numpy


class OBJTriMesh():
    def __init__(self):
        self._clear()

    def _clear(self):
        self.vertices = []
        self.vertex_normals = []
        self.uvs = []
        self.faces = []
        self._adjacency = None
        self._face_normals = {}

    def read(self, filepath):
        self._clear()
        with open(filepath) as handle:
            lines = handle.read().splitlines()
        for line_no, line in enumerate(lines, start=1):
            tokens = line.split()
            if not tokens:
                continue
            tag = tokens[0]
            if tag == "v":
                self.vertices.append((float(tokens[1]), float(tokens[2]), float(tokens[3])))
            elif tag == "vn":
                self.vertex_normals.append((float(tokens[1]), float(tokens[2]), float(tokens[3])))
            elif tag == "vt":
                self.uvs.append((float(tokens[1]), float(tokens[2])))
            elif tag == "f":
                face = self._process_face(line_no, line, tokens)
                if face is not None:
                    self.faces.append(face)
        print(f"Read {len(self.vertices)} vertices, {len(self.vertex_normals)} vertex normals, "
              f"{len(self.uvs)} UVs and {len(self.faces)} faces.")

    def _process_face(self, line_no, line, tokens):
        if len(tokens) != 4:
            print(f"Non triangular face on line {line_no}: {line}")
            return None
        limits = (("vertex", self.vertices), ("uv", self.uvs), ("vertex_normal", self.vertex_normals))
        points = []
        for point_def in tokens[1:]:
            indices = [-1, -1, -1]
            for slot, text in enumerate(point_def.split("/")[:3]):
                if slot > 0 and not text:
                    continue
                value = int(text)
                kind, pool = limits[slot]
                if value > len(pool):
                    print(f"Trying to add out of range {kind} ({value}) on line {line_no} - {line}")
                else:
                    indices[slot] = value - 1
            points.append(tuple(indices))
        return points

    def _face_normal(self, face_index):
        n = self._face_normals.get(face_index)
        if n is None:
            face = self.faces[face_index]
            p = [numpy.array(self.vertices[face[k][0]], dtype=numpy.single) for k in range(3)]
            n = numpy.cross(p[1] - p[0], p[2] - p[0])
            n /= numpy.linalg.norm(n)
            self._face_normals[face_index] = n
        return n

    def get_smooth_vertex_normal(self, vertex_index):
        """Normalised mean of the unit normals of the faces using the vertex
        (one term per use, in face order)."""
        if self._adjacency is None or self._adjacency[0] != len(self.faces):
            uses = {}
            for fi, face in enumerate(self.faces):
                for point in face:
                    uses.setdefault(point[0], []).append(fi)
            self._adjacency = (len(self.faces), uses)
            self._face_normals = {}
        adjacent = numpy.array([self._face_normal(fi) for fi in self._adjacency[1][vertex_index]],
                               dtype=numpy.single)
        average = numpy.mean(adjacent, axis=0)
        average /= numpy.linalg.norm(average)
        return average
