function P = linearTriangulation(p1, p2, M1, M2)
% Drop-in for utils/exe6_8point/linearTriangulation.m.  M1 / M2 may also be 3x4xN (one matrix per
% point), which lets update_landmarks triangulate all its candidates in one call.
assert(size(p1, 1) == 3 && all(size(p1) == size(p2)), 'p1, p2 must be 3xN');
P = vohip_mex('triangulate', double(p1), double(p2), double(M1), double(M2));
end
