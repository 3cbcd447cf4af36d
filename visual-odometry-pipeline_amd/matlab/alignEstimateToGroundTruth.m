function p_G_C = alignEstimateToGroundTruth(pp_G_C, p_V_C)
% Drop-in for utils/exe9_BA/alignEstimateToGroundTruth.m: the 7-parameter similarity fit runs on
% the GPU (Levenberg-Marquardt with analytic Jacobians instead of lsqnonlin).
p_G_C = vohip_mex('align_sim3', double(pp_G_C), double(p_V_C));
end
