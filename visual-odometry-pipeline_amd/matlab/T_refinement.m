function refined_T_W_C = T_refinement(T_W_C, keypoints, landmarks, K)
% Drop-in for utils/T_refinement.m: keypoints 2xN (u, v), landmarks 3xN, at most 100 iterations.
refined_T_W_C = vohip_mex('pose_refine', double(T_W_C), double(keypoints), double(landmarks), K, 100);
end
