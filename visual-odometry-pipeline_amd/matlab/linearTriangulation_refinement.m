function P = linearTriangulation_refinement(P_init, p1, p2, M1, M2)
% Drop-in for utils/linearTriangulation_refinement.m (per-point Levenberg-Marquardt on the GPU
% in place of one lsqnonlin call per point).
P = vohip_mex('triangulate_refine', double(P_init), double(p1), double(p2), double(M1), double(M2));
end
