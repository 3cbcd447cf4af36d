function twist = HomogMatrix2twist(H)
% Drop-in for utils/exe9_BA/HomogMatrix2twist.m: closed-form log, rotation angle in [0, pi].
twist = vohip_mex('homog2twist', double(H));
end
