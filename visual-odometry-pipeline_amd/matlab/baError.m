function error_terms = baError(hidden_state, observations, K)
% Drop-in for utils/exe9_BA/baError.m: 2 x Nobs reprojection errors.
error_terms = vohip_mex('ba_error', double(hidden_state), double(observations), K);
end
