function hidden_state = runBA(hidden_state, observations, K)
% Drop-in for utils/exe9_BA/runBA.m: 21 = the reference's 1 + 20 lsqnonlin iterations.  The
% wall-clock branch of the reference (runBA.m:76) does not exist here: the second stage always runs.
hidden_state = vohip_mex('ba_run', double(hidden_state), double(observations), K, 21);
end
